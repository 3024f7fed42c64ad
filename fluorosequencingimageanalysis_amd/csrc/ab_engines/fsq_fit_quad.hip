// fsq_fit_quad.hip - A/B engine (FSQ_ENGINE_QUAD): the quad-cooperative LM fit (see ../fsq_lm_quad.h) as one persistent launch, in three kernels
//   k3_prep    per candidate: median / max / mean of the 5x5 ROI (start values of pflib.py:199-213)
//   k3_quad    persistent, 4 lanes per fit, work queue with refill: the LM solve (mpfit.py:600-1388)
//   k3_finish  per candidate: fit image, r_2, rmse, illumina_s_n, image coordinates (pflib.py:461-475)
// A/B build only (make ab): the shipped library carries the rounds engine (fsq_fit_rounds.hip) alone.
#ifdef FSQ_BUILD_AB
#include <atomic>

#include "../fsq_common.h"
#include "../fsq_lm_quad.h"

namespace {

// LDS hand-offs between the lanes of one wave: keep the compiler from moving LDS accesses across the point
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ unsigned long long g_quad_queue[256];
#ifdef FSQ_PHASE_PROFILE
__device__ unsigned long long g_phase_cycles[16];
#define PH_DECL unsigned long long ph_t0 = clock64(), ph_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long ph_n_lmpar = 0, ph_n_trips = 0;
#define PH_MARK(k) { unsigned long long t_ = clock64(); ph_acc[k] += t_ - ph_t0; ph_t0 = t_; }
#define PH_FLUSH if (threadIdx.x == 0) { for (int k_ = 0; k_ < 8; k_++) atomicAdd(&g_phase_cycles[k_], ph_acc[k_]); atomicAdd(&g_phase_cycles[10], ph_acc[10]); atomicAdd(&g_phase_cycles[11], ph_acc[11]); atomicAdd(&g_phase_cycles[8], ph_n_lmpar); atomicAdd(&g_phase_cycles[9], ph_n_trips); }
#else
#define PH_DECL
#define PH_MARK(k)
#define PH_FLUSH
#endif

struct QuadOut {            // per candidate, written by k3_quad
    double p[FSQ_NP];
    int status, niter, nfev, pad;
};

template <bool FROM_IMAGE>
FSQ_DEV void load_roi(const uint16_t* __restrict__ src, int H, int W, const int32_t* __restrict__ cand, long long idx,
                      double* data, int* h, int* w, int* field)
{
    if (FROM_IMAGE) {
        *field = cand[3 * idx]; *h = cand[3 * idx + 1]; *w = cand[3 * idx + 2];
        const uint16_t* base = src + ((size_t)*field * H + (*h - 2)) * W + (*w - 2);
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int b = 0; b < 5; b++) data[a * 5 + b] = (double)base[(size_t)a * W + b];
    } else {
        *field = 0; *h = 2; *w = 2;
#pragma unroll
        for (int k = 0; k < FSQ_NPIX; k++) data[k] = (double)src[idx * FSQ_NPIX + k];
    }
}

template <bool FROM_IMAGE>
__global__ void __launch_bounds__(256) k3_prep(const uint16_t* __restrict__ src, int H, int W, const int32_t* __restrict__ cand,
                                               long long n, FsqQuadPrep* __restrict__ prep)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v[FSQ_NPIX];
    int h, w, f;
    load_roi<FROM_IMAGE>(src, H, W, cand, i, v, &h, &w, &f);
    double mx = v[0], isum = 0.0;
#pragma unroll
    for (int k = 0; k < FSQ_NPIX; k++) { mx = v[k] > mx ? v[k] : mx; isum += v[k]; }
    // median = 13th smallest: odd-even transposition sort, fully unrolled (static register indices)
#pragma unroll
    for (int pass = 0; pass < FSQ_NPIX; pass++)
#pragma unroll
        for (int k = (pass & 1); k + 1 < FSQ_NPIX; k += 2) {
            double a = v[k], b = v[k + 1];
            v[k] = a < b ? a : b;
            v[k + 1] = a < b ? b : a;
        }
    prep[i].vmedian = v[12];
    prep[i].vmax = mx;
    prep[i].vmean = isum / 25.0;
}

// ---------------------------------------------------------------------------------------------------
template <bool ALIASED, bool FROM_IMAGE>
__global__ void __launch_bounds__(64, 1) k3_quad(const uint16_t* __restrict__ src, int H, int W,
                                                 const int32_t* __restrict__ cand, long long n,
                                                 const FsqQuadPrep* __restrict__ prep, QuadOut* __restrict__ out,
                                                 unsigned long long* __restrict__ queue)
{
    __shared__ double lds[Q_END * 16];
    const int lane = threadIdx.x, quad = lane >> 2, c4 = lane & 3, qbase = lane & ~3;
    const int n7 = FSQ_NP;
    // per-fit scalars, identical in the 4 lanes of a quad
    long long idx = -1;
    bool active = false, drained = false, fresh = false;
    double llim1 = 0., fnorm = 0., fnorm1 = -1., par = 0., delta = 0., xnorm = 0.;
    int niter = 1, nfev = 0;
    double ca[FSQ_NPIX], cb[FSQ_NPIX], refl[FSQ_NPIX];
    // A quad is either waiting for a fresh Jacobian (need_jqr) or in the middle of mpfit's inner loop with
    // R, qtf, diag held in registers (qlm).  One loop trip = [Jacobian + QR for the quads that need it] +
    // [one lmpar/trial pass for every active quad]: a quad whose step is rejected simply takes another
    // pass on the next trip while its neighbours move on to their next Jacobian.
    bool need_jqr = true;
    unsigned ipvt = 0x76543210u;        // position -> slot
    double gnorm = 0.;
    QuadLm qlm;
    PH_DECL
    for (;;) {
        // ---- refill -------------------------------------------------------------------------------
        if (!active && !drained) {
            unsigned long long got = 0;
            if (c4 == 0) got = atomicAdd(queue, 1ull);
            idx = (long long)__shfl(got, qbase);
            if (idx < n) {
                double d[FSQ_NPIX];
                int h, w, f;
                load_roi<FROM_IMAGE>(src, H, W, cand, idx, d, &h, &w, &f);
#pragma unroll
                for (int k = 0; k < FSQ_NPIX; k++) QL(Q_DATA, k) = d[k];
                const double vmedian = prep[idx].vmedian, vmax = prep[idx].vmax, vmean = prep[idx].vmean;
                llim1 = (vmax - vmean) / 3.0;
                double x0[FSQ_NP] = {vmedian, vmax, 2.5, 2.5, 1., 1., 0.};
#pragma unroll
                for (int i = 0; i < FSQ_NP; i++) {                 // gaussfitter.py:202-204
                    double v = x0[i];
                    if (v > fsq_ulim(i) && fsq_qulim(i)) v = fsq_ulim(i);
                    if (v < fsq_llim(i, llim1)) v = fsq_llim(i, llim1);
                    QL(Q_X, i) = v;
                    QL(Q_DIAG, i) = 0.; QL(Q_SDIAG, i) = 0.;
                }
                niter = 1; nfev = 0; fnorm1 = -1.; par = 0.; delta = 0.; xnorm = 0.;
                active = true; fresh = true; need_jqr = true;
            } else {
                drained = true;
            }
        }
        if (!__any(active)) break;
        PH_MARK(0)
        int status = 0;
        if (active && need_jqr) {
            ipvt = 0x76543210u;
            // ---- fdjac2 (mpfit.py:1512-1612): slot s = column s of the Jacobian, slot 7 = f(x) itself ---
            double xq[FSQ_NP];
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) xq[k] = QL(Q_X, k);
            double hA = 0., hB = 0.;
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                const int slot = c4 + 4 * pass;
                double xp[FSQ_NP];
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) xp[k] = xq[k];
                double hh = 0.;
                if (slot < 7) {
                    double xs = xq[0];
#pragma unroll
                    for (int k = 1; k < FSQ_NP; k++) xs = (slot == k) ? xq[k] : xs;
                    const double eps = 1.4901161193847656e-08;
                    hh = eps * __builtin_fabs(xs);
                    if (hh == 0) hh = eps;
                    double ul = slot < 2 ? 0.0 : slot < 4 ? 3.0 : slot < 6 ? 2.0 : 360.0;
                    if (slot >= 2 && (xs > ul - hh)) hh = -hh;
#pragma unroll
                    for (int k = 0; k < FSQ_NP; k++) xp[k] = (slot == k) ? (xq[k] + hh) : xq[k];
                }
                if (pass == 0) { hA = hh; quad_residual_regs<false>(lds, quad, xp, ca, nullptr, nullptr); }
                else if (slot < 7 || fresh) { hB = hh; quad_residual_regs<false>(lds, quad, xp, cb, nullptr, nullptr); }
            }
            PH_MARK(1)
            if (fresh) {                    // mpfit's first function call (mpfit.py:999): fvec = f(x0)
                if (c4 == 3) {
#pragma unroll
                    for (int i = 0; i < FSQ_NPIX; i++) QL(Q_FVEC, i) = cb[i];
                }
                nfev = 1;
            }
            WAVE_SYNC();
            nfev += 7;
            if (fresh) { fnorm = fsq_sqrt(lds_dot25(lds, quad, Q_FVEC)); fresh = false; }
            // columns: (f(x + h e_j) - fvec) / h ; slot 7 gets a copy of fvec (it becomes Q^T f)
            bool pegA = false, pegB = false;
            {
                double sA = 0.0, sB = 0.0;
#pragma unroll
                for (int i = 0; i < FSQ_NPIX; i++) {
                    double fv = QL(Q_FVEC, i);
                    ca[i] = (ca[i] - fv) / hA;
                    if (c4 < 3) cb[i] = (cb[i] - fv) / hB; else cb[i] = fv;
                    sA += fv * ca[i];
                    sB += fv * cb[i];
                }
                // pegged parameters (mpfit.py:1073-1091): zero the column if the gradient points outward
                {
                    const int slot = c4;
                    double xs = xq[0];
#pragma unroll
                    for (int k = 1; k < FSQ_NP; k++) xs = (slot == k) ? xq[k] : xs;
                    bool lp = (xs == fsq_llim(slot, llim1)), up = fsq_qulim(slot) && (xs == fsq_ulim(slot));
                    pegA = (lp && sA > 0) || (up && sA < 0);
                }
                if (c4 < 3) {
                    const int slot = c4 + 4;
                    double xs = xq[4];
#pragma unroll
                    for (int k = 5; k < FSQ_NP; k++) xs = (slot == k) ? xq[k] : xs;
                    bool lp = (xs == fsq_llim(slot, llim1)), up = (xs == fsq_ulim(slot));
                    pegB = (lp && sB > 0) || (up && sB < 0);
                }
#pragma unroll
                for (int i = 0; i < FSQ_NPIX; i++) {
                    if (pegA) ca[i] = 0;
                    if (pegB) cb[i] = 0;
                }
            }
            PH_MARK(2)
            // ---- qrfac with column pivoting (mpfit.py:1748-1822), Q^T f fused in as slot 7 ----------
            {
                double nA = fsq_sqrt(dot_regcol(ca, 25));
                QL(Q_ACN, c4) = nA; QL(Q_RDIAG, c4) = nA; QL(Q_WA, c4) = nA;
                if (c4 < 3) {
                    double nB = fsq_sqrt(dot_regcol(cb, 25));
                    QL(Q_ACN, c4 + 4) = nB; QL(Q_RDIAG, c4 + 4) = nB; QL(Q_WA, c4 + 4) = nB;
                }
            }
            WAVE_SYNC();
            unsigned pos = 0x76543210u;         // slot -> position
            bool broken = false;
            for (int j = 0; j < n7; j++) {
                const int len = FSQ_NPIX - j;
                if (!broken) {
                    double rmax = QL(Q_RDIAG, j);
                    for (int k = j + 1; k < n7; k++) rmax = np_max2(rmax, QL(Q_RDIAG, k));
                    int kmax = -1;
                    for (int k = n7 - 1; k >= j; k--)
                        if (QL(Q_RDIAG, k) == rmax) kmax = k;
                    if (kmax >= 0 && kmax != j) {
                        int sj = nib_get(ipvt, j), sk = nib_get(ipvt, kmax);
                        ipvt = nib_set(nib_set(ipvt, j, sk), kmax, sj);
                        pos = nib_set(nib_set(pos, sk, j), sj, kmax);
                        QL(Q_RDIAG, kmax) = QL(Q_RDIAG, j);
                        QL(Q_WA, kmax) = QL(Q_WA, j);
                    }
                }
                const int lj = nib_get(ipvt, j);
                const int owner = qbase + (lj & 3);
                const bool useB = (lj >> 2) != 0;
#pragma unroll
                for (int i = 0; i < FSQ_NPIX; i++) refl[i] = quad_bcast(useB ? cb[i] : ca[i], owner);
                double ajj0;
                if (!broken) {
                    double ajnorm = fsq_sqrt(dot_regcol(refl, len));
                    if (ajnorm == 0) broken = true;                 // mpfit.py:1790 `break`
                    else {
                        if (refl[0] < 0) ajnorm = -ajnorm;
#pragma unroll
                        for (int i = 0; i < FSQ_NPIX; i++)
                            if (i < len) refl[i] = refl[i] / ajnorm;
                        refl[0] = refl[0] + 1;
                        QL(Q_TMP, 0) = -ajnorm;                     // rdiag[j] once the step is done
                    }
                }
                ajj0 = refl[0];
                // my two slots: Householder update (remaining J columns) / Q^T f (slot 7, also after a break)
#pragma unroll
                for (int pass = 0; pass < 2; pass++) {
                    const int slot = c4 + 4 * pass;
                    double* col = pass ? cb : ca;
                    const bool is_f = (slot == 7);
                    const int k = is_f ? 7 : nib_get(pos, slot);
                    const bool todo = is_f ? true : (!broken && k > j);
                    if (todo && ajj0 != 0) {
                        double s = 0.0;
#pragma unroll
                        for (int i = 0; i < FSQ_NPIX; i++)
                            if (i < len) s += col[i] * refl[i];
#pragma unroll
                        for (int i = 0; i < FSQ_NPIX; i++)
                            if (i < len) col[i] = col[i] - (refl[i] * s) / ajj0;
                        if (!is_f) {
                            double rk = QL(Q_RDIAG, k);
                            if (rk != 0) {
                                double temp = col[0] / rk;
                                rk = rk * fsq_sqrt(np_max2(1. - fsq_pow2(temp), 0.));
                                temp = rk / QL(Q_WA, k);
                                if ((0.05 * temp * temp) <= FSQ_MACHEP) {
                                    rk = fsq_sqrt(dot_regcol_from1(col, len));
                                    QL(Q_WA, k) = rk;
                                }
                                QL(Q_RDIAG, k) = rk;
                            }
                        }
                    }
                    // row j of R (by slot) / qtf[j]
                    if (is_f) QL(Q_QTF, j) = col[0];
                    else if (slot < 7 && nib_get(pos, slot) > j) QL(Q_R, j * 7 + slot) = col[0];
                }
                if (!broken) QL(Q_RDIAG, j) = QL(Q_TMP, 0);
                QL(Q_R, j * 7 + lj) = QL(Q_RDIAG, j);               // fjac[j, lj] = rdiag[j] (mpfit.py:1123)
                // shift the columns up one row: row j+1 becomes position 0
#pragma unroll
                for (int i = 0; i + 1 < FSQ_NPIX; i++) { ca[i] = ca[i + 1]; cb[i] = cb[i + 1]; }
                WAVE_SYNC();
            }
            PH_MARK(3)
            // ---- first iteration scaling, gradient test (mpfit.py:1099-1160) -----------------------
            if (niter == 1) {
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) {
                    double a = QL(Q_ACN, k);
                    double dg = (a == 0) ? 1. : a;
                    QL(Q_DIAG, k) = dg;
                    QL(Q_WA3, k) = dg * QL(Q_X, k);
                }
                xnorm = fsq_sqrt(lds_dot7(lds, quad, Q_WA3));
                delta = 100. * xnorm;
                if (delta == 0.) delta = 100.;
            }
            PH_MARK(4)
            quadlm_load(qlm, lds, quad, ipvt);
            PH_MARK(10)
            gnorm = 0.;
            if (fnorm != 0) {
#pragma unroll
                for (int j = 0; j < FSQ_NP; j++) {
                    double an = QL(Q_ACN, nib_get(ipvt, j));
                    if (an != 0) {
                        double sg = 0.0;
#pragma unroll
                        for (int i = 0; i < FSQ_NP; i++)
                            if (i <= j) sg += qlm.r[i][j] * qlm.qtf[i];
                        sg = sg / fnorm;
                        gnorm = np_max2(gnorm, __builtin_fabs(sg / an));
                    }
                }
            }
            if (gnorm <= 1e-10) status = 4;
            else {
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) { double dg = QL(Q_DIAG, k), an = QL(Q_ACN, k); dg = (dg > an) ? dg : an; QL(Q_DIAG, k) = dg; qlm.dg[k] = dg; }
#pragma unroll
                for (int j = 0; j < FSQ_NP; j++) qlm.dgp[j] = QL(Q_DIAG, nib_get(ipvt, j));
            }
            need_jqr = false;
            PH_MARK(11)
        }
        PH_MARK(4)
        {
            if (active && status == 0) {
                par = quadlm_lmpar<ALIASED, 16>(qlm, &QL(Q_XLM, 0), ipvt, delta, par);
                PH_MARK(5)
                double wa1[FSQ_NP], wa2[FSQ_NP], xq[FSQ_NP];
                bool lpeg[FSQ_NP], upeg[FSQ_NP];
                int nlpeg = 0, nupeg = 0;
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) {
                    xq[k] = QL(Q_X, k);
                    wa1[k] = -qlm.xp[k];
                    lpeg[k] = (xq[k] == fsq_llim(k, llim1)); nlpeg += lpeg[k];
                    upeg[k] = fsq_qulim(k) && (xq[k] == fsq_ulim(k)); nupeg += upeg[k];
                }
                double alpha = 1.;
                if (nlpeg > 0) {
                    double mxw = wa1[0];
#pragma unroll
                    for (int k = 1; k < FSQ_NP; k++) mxw = np_max2(mxw, wa1[k]);
#pragma unroll
                    for (int k = 0; k < FSQ_NP; k++) if (lpeg[k]) wa1[k] = np_clip(wa1[k], 0., mxw);
                }
                if (nupeg > 0) {
                    double mnw = wa1[0];
#pragma unroll
                    for (int k = 1; k < FSQ_NP; k++) mnw = np_min2(mnw, wa1[k]);
#pragma unroll
                    for (int k = 0; k < FSQ_NP; k++) if (upeg[k]) wa1[k] = np_clip(wa1[k], mnw, 0.);
                }
                {
                    bool any = false; double tmin = 0.;
#pragma unroll
                    for (int k = 0; k < FSQ_NP; k++)
                        if ((__builtin_fabs(wa1[k]) > FSQ_MACHEP) && ((xq[k] + wa1[k]) < fsq_llim(k, llim1))) {
                            double t = (fsq_llim(k, llim1) - xq[k]) / wa1[k];
                            tmin = any ? np_min2(tmin, t) : t; any = true;
                        }
                    if (any) alpha = np_min2(alpha, tmin);
                    any = false;
#pragma unroll
                    for (int k = 0; k < FSQ_NP; k++)
                        if ((__builtin_fabs(wa1[k]) > FSQ_MACHEP) && fsq_qulim(k) && ((xq[k] + wa1[k]) > fsq_ulim(k))) {
                            double t = (fsq_ulim(k) - xq[k]) / wa1[k];
                            tmin = any ? np_min2(tmin, t) : t; any = true;
                        }
                    if (any) alpha = np_min2(alpha, tmin);
                }
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) {
                    wa1[k] = wa1[k] * alpha;
                    wa2[k] = xq[k] + wa1[k];
                    const double ul = fsq_ulim(k), ll = fsq_llim(k, llim1);
                    double sgnu = (ul >= 0) * 2. - 1., sgnl = (ll >= 0) * 2. - 1.;
                    double ulim1 = ul * (1 - sgnu * FSQ_MACHEP) - (ul == 0) * FSQ_MACHEP;
                    double llim1_ = ll * (1 + sgnl * FSQ_MACHEP) + (ll == 0) * FSQ_MACHEP;
                    if (fsq_qulim(k) && (wa2[k] >= ulim1)) wa2[k] = ul;
                    if (wa2[k] <= llim1_) wa2[k] = ll;
                    QL(Q_WA1, k) = wa1[k];
                    QL(Q_WA2, k) = wa2[k];
                }
                double pnorm = 0.0;
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) { double t = QL(Q_DIAG, k) * wa1[k]; pnorm = fsq_fma(t, t, pnorm); }
                pnorm = fsq_sqrt(pnorm);
                if (niter == 1) delta = np_min2(delta, pnorm);
                quad_residual_split(lds, quad, c4, Q_WA2, Q_WA4);
                nfev++;
                fnorm1 = fsq_sqrt(lds_dot25(lds, quad, Q_WA4));
                double actred = -1.;
                if ((0.1 * fnorm1) < fnorm) actred = -fsq_pow2(fnorm1 / fnorm) + 1.;
                double wa3[FSQ_NP];
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) wa3[k] = 0.;
#pragma unroll
                for (int j = 0; j < FSQ_NP; j++) {
                    wa3[j] = 0;
                    double wj = QL(Q_WA1, nib_get(ipvt, j));
#pragma unroll
                    for (int i = 0; i < FSQ_NP; i++)
                        if (i <= j) wa3[i] = wa3[i] + qlm.r[i][j] * wj;
                }
                double t1s = 0.0;
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) { double t = alpha * wa3[k]; t1s = fsq_fma(t, t, t1s); }
                double temp1 = fsq_sqrt(t1s) / fnorm;
                double temp2 = (fsq_sqrt(alpha * par) * pnorm) / fnorm;
                double prered = temp1 * temp1 + (temp2 * temp2) / 0.5;
                double dirder = -(temp1 * temp1 + temp2 * temp2);
                double ratio = 0.;
                if (prered != 0) ratio = actred / prered;
                if (ratio <= 0.25) {
                    double temp;
                    if (actred >= 0) temp = .5;
                    else temp = .5 * dirder / (dirder + .5 * actred);
                    if (((0.1 * fnorm1) >= fnorm) || (temp < 0.1)) temp = 0.1;
                    delta = temp * np_min2(delta, pnorm / 0.1);
                    par = par / temp;
                } else if ((par == 0) || (ratio >= 0.75)) {
                    delta = pnorm / .5;
                    par = .5 * par;
                }
                if (ratio >= 0.0001) {
                    double xs = 0.0;
#pragma unroll
                    for (int k = 0; k < FSQ_NP; k++) {
                        QL(Q_X, k) = wa2[k];
                        double t = QL(Q_DIAG, k) * wa2[k];
                        xs = fsq_fma(t, t, xs);
                    }
                    for (int i = c4; i < FSQ_NPIX; i += 4) QL(Q_FVEC, i) = QL(Q_WA4, i);
                    xnorm = fsq_sqrt(xs);
                    fnorm = fnorm1;
                    niter = niter + 1;
                }
                status = 0;
                bool c1 = (__builtin_fabs(actred) <= 1e-10) && (prered <= 1e-10) && (0.5 * ratio <= 1);
                if (c1) status = 1;
                if (delta <= 1e-10 * xnorm) status = 2;
                if (c1 && (status == 2)) status = 3;
                if (status == 0) {
                    if (niter >= 200) status = 5;
                    if ((__builtin_fabs(actred) <= FSQ_MACHEP) && (prered <= FSQ_MACHEP) && (0.5 * ratio <= 1)) status = 6;
                    if (delta <= FSQ_MACHEP * xnorm) status = 7;
                    if (gnorm <= FSQ_MACHEP) status = 8;
                }
                if (status == 0 && ratio >= 0.0001) need_jqr = true;       // accepted: next trip starts with a new Jacobian
                if (status == 0 && ratio < 0.0001) {
                    bool fin = __builtin_isfinite(ratio);
#pragma unroll
                    for (int k = 0; k < FSQ_NP; k++)
                        fin = fin && __builtin_isfinite(wa1[k]) && __builtin_isfinite(wa2[k]) && __builtin_isfinite(xq[k]);
                    if (!fin) status = -16;
                }
            }
            PH_MARK(6)
#ifdef FSQ_PHASE_PROFILE
            ph_n_lmpar++;
#endif
        }
        // ---- termination ------------------------------------------------------------------------------
        if (active && status != 0) {
            if (c4 == 0) {
                QuadOut o;
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) o.p[k] = QL(Q_X, k);
                o.status = status; o.niter = niter; o.nfev = nfev + (status > 0 ? 1 : 0); o.pad = 0;
                out[idx] = o;
            }
            active = false;
        }
        PH_MARK(7)
#ifdef FSQ_PHASE_PROFILE
        ph_n_trips++;
#endif
    }
    PH_FLUSH
}

// ---------------------------------------------------------------------------------------------------
template <bool FROM_IMAGE>
__global__ void __launch_bounds__(64) k3_finish(const uint16_t* __restrict__ src, int H, int W, const int32_t* __restrict__ cand,
                                                long long n, const FsqQuadPrep* __restrict__ prep,
                                                const QuadOut* __restrict__ qo, FsqRow* __restrict__ rows)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double data[FSQ_NPIX];
    int h, w, f;
    load_roi<FROM_IMAGE>(src, H, W, cand, i, data, &h, &w, &f);
    const QuadOut o = qo[i];
    const double vmax = prep[i].vmax, vmean = prep[i].vmean;
    double fit[FSQ_NPIX];
    fsq_model(o.p, fit);
    double num = 0.0, den = 0.0, rm = 0.0;
    for (int k = 0; k < FSQ_NPIX; k++) { double d = data[k] - fit[k]; num += d * d; }
    for (int k = 0; k < FSQ_NPIX; k++) { double d = data[k] - vmean; den += d * d; }
    for (int k = 0; k < FSQ_NPIX; k++) rm += fsq_pow2(data[k] - fit[k]);
    FsqRow r;
    r.h0 = o.p[2] + h - 2.5;                                            // pflib.py:461
    r.w0 = o.p[3] + w - 2.5;
    r.H = o.p[0]; r.A = o.p[1]; r.sigma_h = o.p[4]; r.sigma_w = o.p[5]; r.theta = o.p[6];
    r.rmse = fsq_sqrt(rm / 25.0);
    r.r2 = 1.0 - num / den;
    {   // pflib.illumina_s_n (pflib.py:261-281)
        double op[16];
        int t = 0;
        for (int ww = 0; ww < 5; ww++) op[t++] = data[ww];
        for (int ww = 0; ww < 5; ww++) op[t++] = data[20 + ww];
        for (int hh = 1; hh < 4; hh++) { op[t++] = data[hh * 5]; op[t++] = data[hh * 5 + 4]; }
        double isum = 0.0;
        for (int k = 0; k < 16; k++) isum += op[k];
        double mean = isum / 16.0, rr[8];
        for (int k = 0; k < 8; k++) { double d0 = op[k] - mean, d1 = op[8 + k] - mean; rr[k] = d0 * d0 + d1 * d1; }
        double res = ((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]));
        res = 0.0 + res;
        r.s_n = (vmax - mean) / fsq_sqrt(res / 16.0);
    }
    r.p2 = o.p[2]; r.p3 = o.p[3];
    r.h = h; r.w = w; r.field = f;
    r.status = o.status; r.niter = o.niter; r.nfev = o.nfev;
    r.key_h = -1; r.key_w = -1;
    rows[i] = r;
}

}  // namespace

// workspace: prep[n] + out[n]
#ifdef FSQ_PHASE_PROFILE
extern "C" int fsq_debug_phase_cycles(unsigned long long* out16, int reset)
{
    unsigned long long* p = nullptr;
    if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_phase_cycles)) != hipSuccess) return -1;
    if (hipMemcpy(out16, p, 16 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (reset) (void)hipMemset(p, 0, 16 * 8);
    return 0;
}
#endif

static int64_t quad_ws_bytes(int64_t n) { return (int64_t)((n + 1) * (sizeof(FsqQuadPrep) + sizeof(QuadOut)) + 512); }

int fsq_launch_fit_quad(const uint16_t* d_src, int H, int W, const int32_t* d_cand, int64_t n, int mode, bool from_image,
                        FsqRow* d_rows, void* d_ws, int64_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < quad_ws_bytes(n) || !d_ws) return FSQ_ENOMEM;
    FsqQuadPrep* prep = (FsqQuadPrep*)d_ws;
    QuadOut* qo = (QuadOut*)(((uintptr_t)(prep + n + 1) + 255) & ~(uintptr_t)255);
    static std::atomic<unsigned> next_slot{0};
    unsigned slot = next_slot.fetch_add(1) % 256u;
    unsigned long long* queue = nullptr;
    FSQ_HIP_CHECK(hipGetSymbolAddress((void**)&queue, HIP_SYMBOL(g_quad_queue)));
    queue += slot;
    FSQ_HIP_CHECK(hipMemsetAsync(queue, 0, sizeof(unsigned long long), s));
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long nq = (n + 15) / 16;
    const long long resident = (long long)cus * 8;
    dim3 gridq((unsigned)(nq < resident ? nq : resident)), blockq(64);
    dim3 gridp((unsigned)((n + 255) / 256)), gridf((unsigned)((n + 63) / 64));
    const bool ref = (mode == FSQ_MODE_REF);
    if (from_image) {
        hipLaunchKernelGGL(k3_prep<true>, gridp, dim3(256), 0, s, d_src, H, W, d_cand, (long long)n, prep);
        if (ref) hipLaunchKernelGGL((k3_quad<true, true>), gridq, blockq, 0, s, d_src, H, W, d_cand, (long long)n, prep, qo, queue);
        else hipLaunchKernelGGL((k3_quad<false, true>), gridq, blockq, 0, s, d_src, H, W, d_cand, (long long)n, prep, qo, queue);
        hipLaunchKernelGGL(k3_finish<true>, gridf, dim3(64), 0, s, d_src, H, W, d_cand, (long long)n, prep, qo, d_rows);
    } else {
        hipLaunchKernelGGL(k3_prep<false>, gridp, dim3(256), 0, s, d_src, H, W, d_cand, (long long)n, prep);
        if (ref) hipLaunchKernelGGL((k3_quad<true, false>), gridq, blockq, 0, s, d_src, H, W, d_cand, (long long)n, prep, qo, queue);
        else hipLaunchKernelGGL((k3_quad<false, false>), gridq, blockq, 0, s, d_src, H, W, d_cand, (long long)n, prep, qo, queue);
        hipLaunchKernelGGL(k3_finish<false>, gridf, dim3(64), 0, s, d_src, H, W, d_cand, (long long)n, prep, qo, d_rows);
    }
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}

#endif  // FSQ_BUILD_AB
