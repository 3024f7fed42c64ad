// fsq_fit_rounds_ka8.h - the Jacobian round with EIGHT lanes per fit (included by fsq_fit_rounds.hip inside its anonymous
// namespace, after kA_jacobian, whose arithmetic it repeats operation for operation).
//
// kA_jacobian gives a fit 4 lanes, each holding two of the 8 columns (7 Jacobian columns + the residual vector): 250
// VGPRs and 19.8 KB of LDS per wave, i.e. 2 waves per SIMD - and the round is latency-bound (PMC: a wave has a VALU
// instruction in flight 36 % of its resident cycles; halving the occupancy costs 1.7x, tools/occupancy_sensitivity.sh).  Here lane c
// of an OCT owns column c alone (slot 7 = f(x), which becomes Q^T f): 8 fits per wave, about half the registers and
// half the LDS, 4 waves per SIMD.  Per fit that is ~10 % more wave-instructions (the pivot bookkeeping and the pivot
// column's scaling are per wave, not per column) against twice the waves to hide the LDS and dependency latencies behind.
// LDS layout: [element][oct], stride 8 (QL below); everything else as in kA_jacobian.
#ifndef FSQ_KA8_SCHED_BARRIER
#define FSQ_KA8_SCHED_BARRIER 1
#endif
// keep the scheduler from interleaving all 25 rows of a column loop (128 registers per lane: five rows in flight are enough)
#ifndef FSQ_KA8_GROUP
#define FSQ_KA8_GROUP 5
#endif
#define KA8_GROUP(i) do { if (FSQ_KA8_GROUP > 0 && ((i) % (FSQ_KA8_GROUP > 0 ? FSQ_KA8_GROUP : 1)) == (FSQ_KA8_GROUP > 0 ? FSQ_KA8_GROUP : 1) - 1) { __builtin_amdgcn_sched_barrier(0); } } while (0)
#pragma push_macro("QL")
#undef QL
#define QL(off, e) lds[((off) + (e)) * 8 + quad]

template <bool FAST>
FSQ_DEV void o8_residual_regs(const double* lds, int quad, const double* p, double* r, int* emin, bool* hz)
{
    bool bad = false;
    double s, c;
    fsq_sincos(FSQ_PI_180 * p[6], &s, &c);
    const double rcen_x = p[3] * c - p[2] * s;
    const double rcen_y = p[3] * s + p[2] * c;
    const FsqDivisor k4 = fsq_divisor(p[4]), k5 = fsq_divisor(p[5]);
    int em = 0;
    if (FAST) {
        *hz = *hz || !fsq_divisor_in_range(p[4]) || !fsq_divisor_in_range(p[5]) || !(__builtin_fabs(p[2]) <= 0x1p100) ||
              !(__builtin_fabs(p[3]) <= 0x1p100);
    }
#pragma unroll
    for (int xi = 0; xi < 5; xi++) {
        // one row of pixels at a time: with 128 registers per lane the 25 table look-ups of exp must not all be in flight
        // at once (the scheduler would otherwise hoist them and spill half the column)
        asm volatile("" ::: "memory");
#if FSQ_KA8_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int yi = 0; yi < 5; yi++) {
            double x = (double)xi, y = (double)yi;
            double xp = x * c - y * s;
            double yp = x * s + y * c;
            double nu = rcen_x - xp, nv = rcen_y - yp;
            if (FAST) { em = min(em, fsq_expo(nu)); em = min(em, fsq_expo(nv)); }
            double u = fsq_div_sel<FAST>(nu, k4);
            double v = fsq_div_sel<FAST>(nv, k5);
            double e = -(u * u + v * v) / 2.;
            double g = p[0] + p[1] * (FAST ? fsq_exp_bf(e, &bad) : fsq_exp(e));
            r[xi * 5 + yi] = QL(Q_DATA, xi * 5 + yi) - g;
        }
        // (pin the running exponent minimum here: left to itself the optimiser turns the chain of mins into one tree at
        // the end of the function and keeps all 50 numerators alive for it)
        if (FAST) asm volatile("" : "+v"(em));
    }
    if (FAST) { *emin = min(*emin, em); *hz = *hz || bad; }
}

FSQ_DEV double o8_dot7(const double* lds, int quad, int off)
{
    double d = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) { double v = QL(off, i); d = fsq_fma(v, v, d); }
    return d;
}
FSQ_DEV double o8_dot25(const double* lds, int quad, int off)
{
    double S[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        double a = QL(off, l), b = QL(off, 4 + l), c = QL(off, 8 + l), e = QL(off, 12 + l);
        S[l] = ((a * a + b * b) + c * c) + e * e;
    }
    double d = (S[0] + S[2]) + (S[1] + S[3]);
#pragma unroll
    for (int i = 16; i < 25; i++) { double v = QL(off, i); d = fsq_fma(v, v, d); }
    return d;
}

#ifndef FSQ_KA8_SCHED_BARRIER
#define FSQ_KA8_SCHED_BARRIER 1
#endif
#ifndef FSQ_KA8_WAVES
#define FSQ_KA8_WAVES 4
#endif
template <bool FAST>
__global__ void __launch_bounds__(64, FSQ_KA8_WAVES) kA8_jacobian(Ctx c, const double* __restrict__ QA, const int* __restrict__ cntA_p,
                                                                 double* __restrict__ QB, int* __restrict__ cntB_p,
                                                                 double* __restrict__ SQ, int* __restrict__ slow_cnt,
                                                                 int* __restrict__ next_counters)
{
    __shared__ double lds[Q_KA_END * 8];
    const int lane = threadIdx.x, quad = lane >> 3, c8 = lane & 7, obase = lane & ~7;      // `quad` = the oct's number
    const int n7 = FSQ_NP;
    if (c.wave_prio) __builtin_amdgcn_s_setprio(3);
    const int cntA = FAST ? *cntA_p : *slow_cnt;
    if (!FAST && blockIdx.x == 0 && threadIdx.x == 0 && cntA > 0) atomicAdd(c.slow_total, cntA);
    if (FAST && blockIdx.x == 0 && threadIdx.x < 4) next_counters[threadIdx.x] = 0;
    const long long cap = c.cap;
    const int base = blockIdx.x * 8;
    if (base >= cntA) return;
    const bool active = (base + quad) < cntA;
    const int qpos = active ? base + quad : 0;
    const double* qa = QA + qpos;
    double col[FSQ_NPIX];
    bool hz = false, qhz = false;
    int emin = 0;
    int idx = 0, tag = 0;
    double llim1 = 0., fnorm = 0., xnorm = 0., delta = 0., par_in = 0.;
    int niter = 1, nfev = 0;
    bool fresh = false;
    unsigned ipvt = 0x76543210u;
    int status = 0;
    double gnorm = 0.;
    if (active) {
        int dummy;
        unpack2(qa[A_IDX * cap], &tag, &dummy);
        unpack2(qa[A_ITER * cap], &niter, &nfev);
        idx = tag_slot(c, tag);
        fresh = (nfev == 0);
        {   // lane c8 converts pixels 4*c8 .. 4*c8+3 of the compact ROI copy (64 bytes per fit)
            const uint2 pw = *(const uint2*)(c.roi + (size_t)idx * 32 + c8 * 4);
            const unsigned w[2] = {pw.x, pw.y};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int k = c8 * 4 + t;
                if (k < FSQ_NPIX) QL(Q_DATA, k) = (double)((w[t >> 1] >> (16 * (t & 1))) & 0xffffu);
            }
        }
        if (!fresh) {
#pragma unroll
            for (int m = 0; m < 4; m++) { const int k = c8 + 8 * m; if (k < FSQ_NPIX) QL(Q_FVEC, k) = c.fvec[(size_t)idx * FSQ_NPIX + k]; }
        }
        if (c8 < FSQ_NP) { QL(Q_X, c8) = qa[(A_X + c8) * cap]; QL(Q_DIAG, c8) = qa[(A_DIAG + c8) * cap]; }
        if (c8 == 7) QL(Q_TMP, 6) = qa[A_LLIM1 * cap];
    }
    WAVE_SYNC();
    if (active) {
        // ---- fdjac2 (mpfit.py:1512-1612): lane s < 7 evaluates f(x + h e_s), lane 7 f(x) itself -----------------------
        double hh = 0.;
        {
            const int slot = c8;
            double xp[FSQ_NP];
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) xp[k] = QL(Q_X, k);
            if (slot < 7) {
                const double xs = QL(Q_X, slot);
                const double eps = 1.4901161193847656e-08;
                hh = eps * __builtin_fabs(xs);
                if (hh == 0) hh = eps;
                double ul = slot < 2 ? 0.0 : slot < 4 ? 3.0 : slot < 6 ? 2.0 : 360.0;
                if (slot >= 2 && (xs > ul - hh)) hh = -hh;
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) xp[k] = (slot == k) ? (xp[k] + hh) : xp[k];
            }
            o8_residual_regs<FAST>(lds, quad, xp, col, &emin, &hz);
        }
        if (fresh) {                    // mpfit's first function call (mpfit.py:999): fvec = f(x0)
            if (c8 == 7) {
#pragma unroll
                for (int i = 0; i < FSQ_NPIX; i++) { QL(Q_FVEC, i) = col[i]; c.fvec[(size_t)idx * FSQ_NPIX + i] = col[i]; }
            }
            nfev = 1;
        }
        WAVE_SYNC();
        nfev += 7;
        if (fresh) QL(Q_TMP, 5) = fsq_sqrt(o8_dot25(lds, quad, Q_FVEC));      // fnorm of a fresh fit
        bool peg = false;
        {
            double sA = 0.0;
            const FsqDivisor kh = fsq_divisor(hh);           // hh = 0 in lane 7: quotient unused
            if (FAST) {
                hz = hz || !fsq_divisor_in_range(hh);
                hz = hz || !(QL(Q_X, 0) <= 0x1p100) || !(QL(Q_X, 1) <= 0x1p100);
            }
#pragma unroll
            for (int i = 0; i < FSQ_NPIX; i++) {
                const double fv = QL(Q_FVEC, i);
                const double nn = col[i] - fv;
                if (FAST) emin = min(emin, fsq_expo(nn));
                const double q = fsq_div_sel<FAST>(nn, kh);
                col[i] = (c8 < 7) ? q : fv;
                sA += fv * col[i];
                if (FAST && (i % 5) == 4) asm volatile("" : "+v"(emin));
                KA8_GROUP(i);
            }
            if (c8 < 7) {   // pegged parameters (mpfit.py:1073-1091)
                const double xs = QL(Q_X, c8), ll = QL(Q_TMP, 6);
                const bool lp = (xs == fsq_llim(c8, ll)), up = fsq_qulim(c8) && (xs == fsq_ulim(c8));
                peg = (lp && sA > 0) || (up && sA < 0);
            }
#pragma unroll
            for (int i = 0; i < FSQ_NPIX; i++)
                if (peg) col[i] = 0;
        }
        // ---- qrfac with column pivoting (mpfit.py:1748-1822), Q^T f fused in as slot 7 -------------------------------
        {
            const double nA = fsq_sqrt(dot_regcol(col, 25));
            if (c8 < 7) { QL(Q_ACN, c8) = nA; QL(Q_RDIAG, c8) = nA; QL(Q_WA, c8) = nA; }
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
            const int owner = obase + lj;
            // the lane that owns the pivot column turns it into the Householder vector and publishes it in LDS (the
            // Q_DATA slots: the pixels are not needed again in this round)
            int emin_s = 0;
            bool brk = false;
            if (lane == owner) {
                double t[FSQ_NPIX];
#pragma unroll
                for (int i = 0; i < FSQ_NPIX; i++) t[i] = col[i];
                if (!broken) {
                    double ajnorm = fsq_sqrt(dot_regcol(t, len));
                    if (ajnorm == 0) brk = true;                // mpfit.py:1790 `break`
                    else {
                        if (t[0] < 0) ajnorm = -ajnorm;
                        const FsqDivisor kn = fsq_divisor(ajnorm);
                        if (FAST) {
                            int er = 0;
#pragma unroll
                            for (int i = 0; i < FSQ_NPIX; i++) er = min(er, fsq_expo(t[i]));
                            emin = min(emin, er);
                            hz = hz || !fsq_divisor_in_range(ajnorm);
                            emin_s = er - fsq_expo(ajnorm) - 1;
                        }
#pragma unroll
                        for (int i = 0; i < FSQ_NPIX; i++) { t[i] = fsq_div_sel<FAST>(t[i], kn); KA8_GROUP(i); }   // rows >= len are zeros
                        t[0] = t[0] + 1;
                        QL(Q_TMP, 0) = -ajnorm;
                    }
                }
#pragma unroll
                for (int i = 0; i < FSQ_NPIX; i++) QL(Q_DATA, i) = t[i];
            }
            WAVE_SYNC();
            broken = broken || (__shfl((int)brk, owner) != 0);
            emin_s = __shfl(emin_s, owner);
            const double ajj0 = QL(Q_DATA, 0);
            const FsqDivisor kj = fsq_divisor(ajj0);
            if (FAST) hz = hz || !fsq_divisor_in_range(ajj0);
            {
                const int slot = c8;
                const bool is_f = (slot == 7);
                const int k = is_f ? 7 : nib_get(pos, slot);
                const bool todo = is_f ? true : (!broken && k > j);
                if (todo && ajj0 != 0) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < FSQ_NPIX; i++) s += col[i] * QL(Q_DATA, i);
                    if (FAST) {
                        const int es = fsq_expo(s);
                        hz = hz || (emin_s + es - 2 < -FSQ_DIV_EN) || (es + 2 > FSQ_DIV_EN);
                    }
#pragma unroll
                    for (int i = 0; i < FSQ_NPIX; i++) { col[i] = col[i] - fsq_div_sel<FAST>(QL(Q_DATA, i) * s, kj); KA8_GROUP(i); }
                }
                if (is_f) QL(Q_QTF, j) = col[0];
                else if (nib_get(pos, slot) > j) QL(Q_R, j * 7 + slot) = col[0];
            }
            // norm down-dating of the live columns (mpfit.py:1810-1820): lane c8 takes position j + 1 + c8
            WAVE_SYNC();
            unsigned long long redo = 0ull;
            {
                const int p = j + 1 + c8;
                bool need = false;
                if (p < n7 && !broken && ajj0 != 0) {
                    double rk = QL(Q_RDIAG, p);
                    if (rk != 0) {
                        double temp = QL(Q_R, j * 7 + nib_get(ipvt, p)) / rk;
                        rk = rk * fsq_sqrt(np_max2(1. - fsq_pow2(temp), 0.));
                        temp = rk / QL(Q_WA, p);
                        if ((0.05 * temp * temp) <= FSQ_MACHEP || c.force_redo) need = true;
                        else QL(Q_RDIAG, p) = rk;
                    }
                }
                redo = __ballot(need);
            }
            if (redo) {
                const int slot = c8;
                const int k = (slot < 7) ? nib_get(pos, slot) : 0;
                const int ix = k - j - 1;
                if (slot < 7 && ix >= 0 && ((redo >> (obase + ix)) & 1ull)) {
                    const double rk = fsq_sqrt(dot_regcol_from1(col, len));
                    QL(Q_WA, k) = rk;
                    QL(Q_RDIAG, k) = rk;
                }
            }
            if (!broken) QL(Q_RDIAG, j) = QL(Q_TMP, 0);
            QL(Q_R, j * 7 + lj) = QL(Q_RDIAG, j);               // fjac[j, lj] = rdiag[j] (mpfit.py:1123)
#pragma unroll
            for (int i = 0; i + 1 < FSQ_NPIX; i++) col[i] = col[i + 1];
            col[FSQ_NPIX - 1] = 0.0;
            WAVE_SYNC();
        }
        // ---- first iteration scaling, gradient test (mpfit.py:1099-1160) ---------------------------------------------
        llim1 = qa[A_LLIM1 * cap]; par_in = qa[A_PAR * cap]; delta = qa[A_DELTA * cap]; xnorm = qa[A_XNORM * cap];
        fnorm = fresh ? QL(Q_TMP, 5) : qa[A_FNORM * cap];
        if (niter == 1) {
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) {
                double a = QL(Q_ACN, k);
                double dg = (a == 0) ? 1. : a;
                QL(Q_DIAG, k) = dg;
                QL(Q_WA3, k) = dg * QL(Q_X, k);
            }
            xnorm = fsq_sqrt(o8_dot7(lds, quad, Q_WA3));
            delta = 100. * xnorm;
            if (delta == 0.) delta = 100.;
        }
        gnorm = 0.;
        if (fnorm != 0) {
            for (int j = 0; j < n7; j++) {
                double an = QL(Q_ACN, nib_get(ipvt, j));
                if (an != 0) {
                    double sg = 0.0;
                    for (int i = 0; i <= j; i++) sg += QR(i, j) * QL(Q_QTF, i);
                    sg = sg / fnorm;
                    gnorm = np_max2(gnorm, __builtin_fabs(sg / an));
                }
            }
        }
        if (gnorm <= 1e-10) status = 4;                                    // mpfit.py:1151
        else {
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) { double dg = QL(Q_DIAG, k), an = QL(Q_ACN, k); QL(Q_DIAG, k) = (dg > an) ? dg : an; }
        }
        if (FAST) {
            hz = hz || (emin < -FSQ_DIV_EN);
            if (c.force_slow_mod > 0 && (idx % c.force_slow_mod) == 0) hz = true;
            const unsigned long long m = __ballot(hz);
            qhz = ((m >> obase) & 0xffull) != 0;
        }
        if (status != 0 && c8 == 0 && !qhz) {
            FitOut o;
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) o.x[k] = QL(Q_X, k);
            o.status = status; o.niter = niter; o.nfev = nfev; o.pad = 0;
            c.out[idx] = o;
        }
    }
    wave_mark_done(c.done, active && status != 0 && c8 == 0 && !qhz, tag_ticket(c, tag));
    // ---- ... or hand it over to the step round: one queue-B slot per surviving oct ------------------------------------
    {
        bool go = active && (status == 0);
        if (FAST) {
            int sat = wave_reserve(slow_cnt, qhz && c8 == 0);
            sat = __shfl(sat, obase);
            if (qhz) for (int f = c8; f < A_LEN; f += 8) SQ[(size_t)sat + f * cap] = qa[f * cap];
            go = go && !qhz;
        }
        int at = wave_reserve(cntB_p, go && c8 == 0);
        at = __shfl(at, obase);
        if (go) {
            double* qb = QB + at;
            for (int e = c8; e < 28; e += 8) {
                int i = 0, rem = e;
                while (rem >= 7 - i) { rem -= 7 - i; i++; }
                qb[(B_R + e) * cap] = QR(i, i + rem);
            }
            if (c8 < FSQ_NP) {
                const int k = c8;
                qb[(A_X + k) * cap] = QL(Q_X, k);
                qb[(A_DIAG + k) * cap] = QL(Q_DIAG, k);
                qb[(B_QTF + k) * cap] = QL(Q_QTF, k);
                qb[(B_SDIAG + k) * cap] = 0.;
            }
            if (c8 == 7) {
                qb[A_IDX * cap] = pack2(tag, 0);
                qb[A_LLIM1 * cap] = llim1; qb[A_FNORM * cap] = fnorm; qb[A_PAR * cap] = par_in; qb[A_DELTA * cap] = delta;
                qb[A_XNORM * cap] = xnorm; qb[A_ITER * cap] = pack2(niter, nfev);
                qb[B_GNORM * cap] = gnorm; qb[B_IPVT * cap] = pack2((int)ipvt, 0);
            }
        }
    }
}
#pragma pop_macro("QL")
#undef KA8_GROUP
