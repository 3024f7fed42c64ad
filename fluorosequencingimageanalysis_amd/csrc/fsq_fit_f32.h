// fsq_fit_f32.h - FSQ_MODE_TEXTBOOK_F32: the PSF fit as a plain single-precision Levenberg-Marquardt solve.
// (included by fsq_fit_rounds.hip inside its anonymous namespace: shares Ctx / BatchArgs / kinit / kfinish)
//
// BASELINE configs[4] asks for "fp16 pixel loads / fp32 LM accumulate".  The reference-faithful fit cannot run in
// fp32 (its forward-difference step 1.5e-8 |x| is below fp32 resolution, mpfit.py:1529,1557, and its result is chaotic at
// the 1-ulp level), so this mode is an OPT-IN approximation and never the default: the same model
// (gaussfitter.twodgaussian, agpy/gaussfitter.py:100-136), the same start, bounds and pegging rule as pflib / mpfit
// (pflib.py:199-212, mpfit.py:1073-1091, 1173-1233), but an analytic Jacobian, 7x7 normal equations accumulated in fp32
// registers, a Cholesky solve and Marquardt damping (Nielsen's update) instead of MINPACK's QR + lmpar trust region.
// Agreement with the fp64 textbook solver is REPORTED (bench.py extras), not asserted: about 70 % of the
// kept fits agree to 1e-4 and 82 % to 1e-3; the same algorithm in fp64 reaches 81 % (the fit itself moves by more than
// 1e-4 under 1-ulp noise for a fifth of the kept fits, SURVEY PROBE 8).
//
// Execution: one lane per fit, everything in registers, and LANE-LEVEL REFILL - every trip of the loop is one LM trial
// (solve, bounded step, model + Jacobian + normal equations at the trial point, accept / reject) for every lane, and a lane
// whose fit has terminated takes the next candidate from a counter (one wave-aggregated atomic per trip).  All lanes run
// the same straight-line trip whatever their iteration counts are, so the spread of 3...200 iterations per fit costs
// nothing; a wave leaves when the counter is exhausted and its lanes are idle.
#pragma once

#define F32_FTOL 3e-7f
#define F32_XTOL 1e-6f
#define F32_GTOL 1e-7f
#define F32_LAMBDA0 1e-2f
#define F32_MAX_ITER 200
#define F32_MAX_TRIPS 400

FSQ_DEV int f32_tri(int i, int j) { return i * (i + 1) / 2 + j; }      // (i, j), i >= j, of a packed lower triangle

FSQ_DEV float f32_lo(int k, float llim1) { return k == 0 ? 0.f : k == 1 ? llim1 : k < 4 ? 2.f : k < 6 ? 0.75f : 0.f; }
FSQ_DEV float f32_hi(int k) { return k < 4 ? 3.f : k < 6 ? 2.f : 360.f; }           // (k >= 2 only)

// model, Jacobian and normal equations at x: chi2 = sum r^2, N = J^T J (packed lower triangle), g = J^T r, r = model - data.
// The pixels sit in the lane's LDS column (pix[k * 64]); the loop over them is ROLLED - unrolled, the compiler evaluates
// many pixels side by side and spills hundreds of registers; 35 accumulators and one pixel's terms is all that has to live.
FSQ_DEV void f32_eval(const float* x, const float* pix, float& chi2_out, float* N, float* g)
{
    float sn, cs;
    sincosf(x[6] * 0.017453292519943295f, &sn, &cs);
    const float i4 = __builtin_amdgcn_rcpf(x[4]), i5 = __builtin_amdgcn_rcpf(x[5]);
    float chi2 = 0.f;
#pragma unroll
    for (int k = 0; k < 28; k++) N[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 7; k++) g[k] = 0.f;
    float A = x[3];                                 // x[3] - row
#pragma unroll 1
    for (int row = 0; row < 5; row++) {
        float B = x[2];                             // x[2] - column
#pragma unroll 1
        for (int col = 0; col < 5; col++) {
            const float d = pix[(row * 5 + col) * 64];
            const float nu = A * cs - B * sn, nv = A * sn + B * cs;
            const float u = nu * i4, v = nv * i5;
            const float E = __expf(-0.5f * (u * u + v * v));
            const float aE = x[1] * E;
            const float r = (x[0] - d) + aE;
            float J[7];
            J[0] = 1.f;
            J[1] = E;
            J[2] = aE * (u * sn * i4 - v * cs * i5);
            J[3] = -aE * (u * cs * i4 + v * sn * i5);
            J[4] = aE * u * u * i4;
            J[5] = aE * v * v * i5;
            J[6] = aE * (u * nv * i4 - v * nu * i5) * 0.017453292519943295f;
            chi2 = fmaf(r, r, chi2);
#pragma unroll
            for (int a = 0; a < 7; a++) {
                g[a] = fmaf(J[a], r, g[a]);
#pragma unroll
                for (int b_ = 0; b_ <= a; b_++) N[f32_tri(a, b_)] = fmaf(J[a], J[b_], N[f32_tri(a, b_)]);
            }
            B -= 1.f;
        }
        A -= 1.f;
    }
    chi2_out = chi2;
}

__global__ void __launch_bounds__(64, 3) kfit_f32(Ctx c, BatchArgs b, int* __restrict__ next)
{
    int idx = -1, status = 0, niter = 1, nfev = 0;
    long long slot = 0;
    bool exhausted = false, first = true;
    float x[7], D[7], N[28], g[7], chi2 = 0.f, lam = F32_LAMBDA0, nu = 2.f, llim1 = 0.f;
    __shared__ float pixels[FSQ_NPIX * 64];
    float* pix = pixels + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 7; k++) { x[k] = 1.f; D[k] = 1.f; g[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 28; k++) N[k] = 0.f;
    for (;;) {
        const bool need = (idx < 0) && !exhausted;
        const int pos = wave_reserve(next, need);
        if (need) {
            if (pos < b.n) {
                idx = pos; slot = b.base + pos;
                const uint16_t* src = c.roi + (size_t)slot * 32;
                const uint4 q0 = *(const uint4*)src, q1 = *(const uint4*)(src + 8), q2 = *(const uint4*)(src + 16);
                const unsigned w[13] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, *(const unsigned*)(src + 24)};
#pragma unroll
                for (int k = 0; k < FSQ_NPIX; k++) pix[k * 64] = (float)((w[k >> 1] >> (16 * (k & 1))) & 0xffffu);
#pragma unroll
                for (int k = 0; k < 7; k++) { x[k] = (float)c.out[slot].x[k]; D[k] = 0.f; g[k] = 0.f; }   // the clipped start (kinit)
#pragma unroll
                for (int k = 0; k < 28; k++) N[k] = 0.f;
                llim1 = (float)((c.stat[slot].vmax - c.stat[slot].vmean) / 3.0);
                x[1] = fmaxf(x[1], llim1);                  // (the bound as this precision sees it)
                chi2 = 0.f; lam = F32_LAMBDA0; nu = 2.f; niter = 1; nfev = 0; status = 0; first = true;
            } else exhausted = true;
        }
        if (__ballot(idx >= 0) == 0ull) break;
        if (idx >= 0) {
            // ---- the step: (N + lam D^2) p = -g over the parameters that are not pegged at a bound ------------------
            float p[7];
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 7; k++) p[k] = 0.f;
            if (!first) {
                bool peg[7];
#pragma unroll
                for (int k = 0; k < 7; k++)
                    peg[k] = (x[k] <= f32_lo(k, llim1) && g[k] > 0.f) || (k >= 2 && x[k] >= f32_hi(k) && g[k] < 0.f);
                float a[28], y[7];
#pragma unroll
                for (int i = 0; i < 7; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) {
                        float v = N[f32_tri(i, j)];
                        if (i == j) v = fmaf(lam * D[i], D[i], v);
                        if (peg[i] || peg[j]) v = (i == j) ? 1.f : 0.f;
                        a[f32_tri(i, j)] = v;
                    }
#pragma unroll
                for (int j = 0; j < 7; j++) {               // Cholesky, in place: a = L, the diagonal holds 1 / L(j, j)
                    float s = a[f32_tri(j, j)];
#pragma unroll
                    for (int k = 0; k < j; k++) s = fmaf(-a[f32_tri(j, k)], a[f32_tri(j, k)], s);
                    ok = ok && (s > 0.f);
                    const float inv = __builtin_amdgcn_rsqf(s > 0.f ? s : 1.f);
                    a[f32_tri(j, j)] = inv;
#pragma unroll
                    for (int i = j + 1; i < 7; i++) {
                        float t = a[f32_tri(i, j)];
#pragma unroll
                        for (int k = 0; k < j; k++) t = fmaf(-a[f32_tri(i, k)], a[f32_tri(j, k)], t);
                        a[f32_tri(i, j)] = t * inv;
                    }
                }
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    float t = peg[i] ? 0.f : -g[i];
#pragma unroll
                    for (int k = 0; k < i; k++) t = fmaf(-a[f32_tri(i, k)], y[k], t);
                    y[i] = t * a[f32_tri(i, i)];
                }
#pragma unroll
                for (int i = 6; i >= 0; i--) {
                    float t = y[i];
#pragma unroll
                    for (int k = i + 1; k < 7; k++) t = fmaf(-a[f32_tri(k, i)], p[k], t);
                    p[i] = t * a[f32_tri(i, i)];
                }
            }
            // ---- keep the step inside the bounds: the whole step is shortened (mpfit.py:1192-1216), then snapped -----
            float alpha = 1.f;
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const float lo = f32_lo(k, llim1);
                if (p[k] < 0.f && x[k] + p[k] < lo) alpha = fminf(alpha, (lo - x[k]) / p[k]);
                if (k >= 2 && p[k] > 0.f && x[k] + p[k] > f32_hi(k)) alpha = fminf(alpha, (f32_hi(k) - x[k]) / p[k]);
            }
            float xt[7], ps[7];
#pragma unroll
            for (int k = 0; k < 7; k++) {
                ps[k] = alpha * p[k];
                float t = x[k] + ps[k];
                const float lo = f32_lo(k, llim1);
                if (t <= lo * (1.f + 1.1920929e-7f)) t = lo;
                if (k >= 2 && t >= f32_hi(k) * (1.f - 1.1920929e-7f)) t = f32_hi(k);
                xt[k] = t;
            }
            // ---- the trial point: residuals, Jacobian, normal equations in one pass ----------------------------------
            float chi2t, Nt[28], gt[7];
            f32_eval(xt, pix, chi2t, Nt, gt);
            nfev++;
            float gp = 0.f, pNp = 0.f;
#pragma unroll
            for (int i = 0; i < 7; i++) {
                gp = fmaf(g[i], ps[i], gp);
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < 7; j++) t = fmaf(N[i >= j ? f32_tri(i, j) : f32_tri(j, i)], ps[j], t);
                pNp = fmaf(t, ps[i], pNp);
            }
            const float pred = -(2.f * gp + pNp);
            const float rho = (pred > 0.f) ? (chi2 - chi2t) / pred : -1.f;
            const bool finite_t = (chi2t == chi2t) && (chi2t < 3.0e38f);
            const bool acc = first ? finite_t : (ok && finite_t && rho > 1e-4f);
            if (acc) {
                if (!first) {
                    const float actred = (chi2 > 0.f) ? (chi2 - chi2t) / chi2 : 0.f, prered = (chi2 > 0.f) ? pred / chi2 : 0.f;
                    float dxn = 0.f, xn = 0.f;
#pragma unroll
                    for (int k = 0; k < 7; k++) { const float a_ = D[k] * ps[k], b_ = D[k] * xt[k]; dxn = fmaf(a_, a_, dxn); xn = fmaf(b_, b_, xn); }
                    if (fabsf(actred) <= F32_FTOL && prered <= F32_FTOL) status = 1;
                    else if (dxn <= F32_XTOL * F32_XTOL * xn) status = 2;
                    const float f = 2.f * rho - 1.f;
                    lam = lam * fmaxf(1.f / 3.f, 1.f - f * f * f);
                    nu = 2.f;
                    niter++;
                }
                chi2 = chi2t;
#pragma unroll
                for (int k = 0; k < 7; k++) { x[k] = xt[k]; g[k] = gt[k]; }
#pragma unroll
                for (int k = 0; k < 28; k++) N[k] = Nt[k];
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    const float dn = __builtin_sqrtf(N[f32_tri(k, k)]);
                    D[k] = fmaxf(D[k], dn);
                    if (D[k] == 0.f) D[k] = 1.f;
                }
                if (chi2 == 0.f) status = 1;                 // an exact fit (flat ROIs)
                else if (status == 0) {
                    float gn = 0.f;
                    const float fn = __builtin_sqrtf(chi2);
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        const bool pegk = (x[k] <= f32_lo(k, llim1) && g[k] > 0.f) || (k >= 2 && x[k] >= f32_hi(k) && g[k] < 0.f);
                        if (!pegk) gn = fmaxf(gn, fabsf(g[k]) / (D[k] * fn));
                    }
                    if (gn <= F32_GTOL) status = 4;
                }
            } else if (!finite_t && first) status = -16;     // (cannot happen with finite pixels: kept for the row's sake)
            else {
                lam = lam * nu;
                nu = nu * 2.f;
                if (lam > 1e10f) status = 2;                 // no admissible step left: the point is as good as this precision gets
            }
            if (status == 0 && (niter >= F32_MAX_ITER || nfev >= F32_MAX_TRIPS)) status = 5;
            first = false;
            if (status != 0) {
                FitOut o;
#pragma unroll
                for (int k = 0; k < 7; k++) o.x[k] = (double)x[k];
                o.status = status; o.niter = niter; o.nfev = nfev; o.pad = 0;
                c.out[slot] = o;
                idx = -1;
            }
        }
    }
}
