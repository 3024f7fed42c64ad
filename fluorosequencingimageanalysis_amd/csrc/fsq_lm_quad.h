// fsq_lm_quad.h - the LM PSF fit mapped onto a QUAD of lanes (4 lanes per fit, 16 fits per wave).
//
// Same arithmetic, in the same order, as fsq_lm_core.h (and therefore as the reference's
// mpfit/gaussfitter path); what changes is where the work and the data live on a CDNA4 CU:
//   * the 25x7 Jacobian plus the residual vector are 8 columns; lane c of a quad owns columns c and
//     c+4 in REGISTERS (2 x 25 doubles) - fdjac2's seven model evaluations, the column norms, the
//     Householder updates and Q^T f all run column-parallel with no cross-lane sums, because every
//     python sum() of the reference runs down a column;
//   * the pivot column (the reflector) is handed round the quad each step - with ds_bpermute in the persistent
//     engine (fsq_fit_quad.hip), through LDS by the lane that owns it in the rounds engine (fsq_fit_rounds.hip) -
//     and the column registers shift up one row per step, so all register indices stay static;
//   * the 7x7 part (R, lmpar, qrsolv, the trust-region bookkeeping) lives in LDS, one 8-byte slot per
//     quad per element ([element][quad] layout: the 4 lanes of a quad broadcast-read the same address,
//     different quads hit different banks) and is executed redundantly by the 4 lanes;
//   * persistent engine: a wave advances all its 16 fits by one outer LM iteration per loop trip and refills
//     finished quads from a global queue, so the 1..200 iteration spread costs no idle lanes; rounds engine: only
//     the Jacobian round runs in this layout, the 7x7 part runs one lane per fit (QuadLm below).
#pragma once
#include "fsq_lm_core.h"

// ---- LDS layout (units: doubles, per quad; element e of quad q lives at lds[(e)*16 + q]) ----------
enum {
    // used by the Jacobian round (first Q_KA_END doubles only)
    Q_X = 0, Q_DIAG = 7, Q_TMP = 14 /* 7 */, Q_QTF = 21, Q_WA3 = 28, Q_ACN = 35 /* acnorm by slot */,
    Q_RDIAG = 42 /* by logical position */, Q_WA = 49 /* by logical position */,
    Q_R = 56 /* 7 rows x 7 slots: R(i,k) = Q_R + i*7 + slot(k) */, Q_FVEC = 105, Q_DATA = 130, Q_KA_END = 155,
    // the Jacobian round stages the model evaluations of its pixel-split fdjac2 in the 84 doubles from Q_QTF to the end of
    // Q_R, which are only written later (qrfac) - three 25-pixel columns at a time
    Q_STAGE = 21,
    // additionally used by the single-launch persistent quad engine
    Q_SDIAG = 155, Q_XLM = 162, Q_WA1 = 169, Q_WA2 = 176, Q_WA4 = 183, Q_TMP2 = 208 /* 7 */, Q_TMP3 = 215 /* 7 */, Q_END = 222
};
#define QL(off, e) lds[((off) + (e)) * 16 + quad]

struct FsqQuadPrep {       // per candidate, written by the prep kernel
    double vmedian, vmax, vmean;
};

// packed 4-bit lists (position -> slot and slot -> position)
FSQ_DEV int nib_get(unsigned w, int k) { return (int)((w >> (4 * k)) & 15u); }
FSQ_DEV unsigned nib_set(unsigned w, int k, int v) { return (w & ~(15u << (4 * k))) | ((unsigned)v << (4 * k)); }

FSQ_DEV double quad_bcast(double v, int src_lane) { return __shfl(v, src_lane); }

// OpenBLAS strided ddot pattern on a register column of `len` valid rows (positions 0..len-1), 19 <= len <= 25
// (the qrfac of a 25x7 system).  Branch-free: the group-of-4 / scalar-tail decision of the BLAS kernel is made
// with selects, so the call sits inside one basic block.
FSQ_DEV double dot_regcol(const double* x, int len)
{
    __builtin_assume(len >= 19 && len <= 25);
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int s = 0; s < 7; s++) {
        if (4 * s + 3 < 19) {
            t2 += fsq_fma(x[4 * s + 1], x[4 * s + 1], x[4 * s + 3] * x[4 * s + 3]);
            t1 += fsq_fma(x[4 * s], x[4 * s], x[4 * s + 2] * x[4 * s + 2]);
        } else {
            const bool full = (4 * s + 3 < len);
            double g1 = t1;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * s + r < 25) { double f = fsq_fma(x[4 * s + r], x[4 * s + r], g1); g1 = (4 * s + r < len) ? f : g1; }
            if (4 * s + 3 < 25) {
                double f2 = t2 + fsq_fma(x[4 * s + 1], x[4 * s + 1], x[4 * s + 3] * x[4 * s + 3]);
                double f1 = t1 + fsq_fma(x[4 * s], x[4 * s], x[4 * s + 2] * x[4 * s + 2]);
                t2 = full ? f2 : t2;
                t1 = full ? f1 : g1;
            } else t1 = g1;
        }
    }
    return t2 + t1;
}
// same pattern starting one row lower (enorm(a[j+1:, lk]) of the norm re-computation)
FSQ_DEV double dot_regcol_from1(const double* x, int len)
{
    __builtin_assume(len >= 19 && len <= 25);
    double t1 = 0.0, t2 = 0.0;
    const int n = len - 1;
#pragma unroll
    for (int s = 0; s < 6; s++) {
        if (4 * s + 3 < 18) {
            t2 += fsq_fma(x[4 * s + 2], x[4 * s + 2], x[4 * s + 4] * x[4 * s + 4]);
            t1 += fsq_fma(x[4 * s + 1], x[4 * s + 1], x[4 * s + 3] * x[4 * s + 3]);
        } else {
            const bool full = (4 * s + 3 < n);
            double g1 = t1;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * s + r + 1 < 25) { double f = fsq_fma(x[4 * s + r + 1], x[4 * s + r + 1], g1); g1 = (4 * s + r < n) ? f : g1; }
            if (4 * s + 4 < 25) {
                double f2 = t2 + fsq_fma(x[4 * s + 2], x[4 * s + 2], x[4 * s + 4] * x[4 * s + 4]);
                double f1 = t1 + fsq_fma(x[4 * s + 1], x[4 * s + 1], x[4 * s + 3] * x[4 * s + 3]);
                t2 = full ? f2 : t2;
                t1 = full ? f1 : g1;
            } else t1 = g1;
        }
    }
    return t2 + t1;
}

// model residuals data - g at parameters p, all 25 pixels into registers (data read from LDS).
// FAST: the two divisions per pixel share their divisors (sigma_h, sigma_w) -> fsq_div_by, exp is the branch-free
// fsq_exp_bf; *emin collects the smallest numerator exponent and *hz is raised when a divisor, the centre or an exp
// argument leaves the range those are exact in (the caller then repeats the fit with FAST = false).
template <bool FAST>
FSQ_DEV void quad_residual_regs(const double* lds, int quad, const double* p, double* r, int* emin, bool* hz)
{
    bool bad = false;
    double s, c;
    fsq_sincos(FSQ_PI_180 * p[6], &s, &c);
    const double rcen_x = p[3] * c - p[2] * s;
    const double rcen_y = p[3] * s + p[2] * c;
    const FsqDivisor k4 = fsq_divisor(p[4]), k5 = fsq_divisor(p[5]);
    int em = 0;
    if (FAST) {
        // |numerator| <= |p2| + |p3| + 8: bounded once the centre is
        *hz = *hz || !fsq_divisor_in_range(p[4]) || !fsq_divisor_in_range(p[5]) || !(__builtin_fabs(p[2]) <= 0x1p100) ||
              !(__builtin_fabs(p[3]) <= 0x1p100);
    }
#pragma unroll
    for (int xi = 0; xi < 5; xi++)
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
    if (FAST) { *emin = min(*emin, em); *hz = *hz || bad; }
}

// trial-point residuals: the 25 pixels are split over the 4 lanes of the quad, results go to LDS
FSQ_DEV void quad_residual_split(double* lds, int quad, int c4, int p_off, int out_off)
{
    double p[FSQ_NP];
#pragma unroll
    for (int k = 0; k < FSQ_NP; k++) p[k] = QL(p_off, k);
    double s, c;
    fsq_sincos(FSQ_PI_180 * p[6], &s, &c);
    const double rcen_x = p[3] * c - p[2] * s;
    const double rcen_y = p[3] * s + p[2] * c;
    for (int i = c4; i < FSQ_NPIX; i += 4) {
        int xi = i / 5, yi = i - 5 * xi;
        double x = (double)xi, y = (double)yi;
        double xp = x * c - y * s;
        double yp = x * s + y * c;
        double u = (rcen_x - xp) / p[4];
        double v = (rcen_y - yp) / p[5];
        double e = -(u * u + v * v) / 2.;
        double g = p[0] + p[1] * fsq_exp(e);
        QL(out_off, i) = QL(Q_DATA, i) - g;
    }
}

FSQ_DEV double lds_dot7(const double* lds, int quad, int off)
{
    double d = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) { double v = QL(off, i); d = fsq_fma(v, v, d); }
    return d;
}
FSQ_DEV double lds_dot25(const double* lds, int quad, int off)
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

// R(i,k) with k a LOGICAL column: stored by slot
#define QR(i, k) QL(Q_R, (i) * 7 + nib_get(ipvt, (k)))

// mpfit.qrsolv on the LDS-resident R (mpfit.py:1903-1978); solution -> Q_XLM (by parameter index).
// Loops are rolled on purpose (code size); wa lives in Q_TMP2, the saved diagonal (textbook mode) in Q_TMP3.
template <bool ALIASED>
FSQ_DEV void quad_qrsolv(double* lds, int quad, unsigned ipvt, int d_off /* sqrt(par)*diag, by parameter */)
{
    const int n = FSQ_NP;
    for (int j = 0; j < n; j++)
        for (int i = j; i < n; i++) QR(i, j) = QR(j, i);
    for (int j = 0; j < n; j++) { if (!ALIASED) QL(Q_TMP3, j) = QR(j, j); QL(Q_TMP2, j) = QL(Q_QTF, j); }
    for (int j = 0; j < n; j++) {
        int l = nib_get(ipvt, j);
        double dl = QL(d_off, l);
        if (dl == 0) break;
        for (int k = j; k < n; k++) QL(Q_SDIAG, k) = 0;
        QL(Q_SDIAG, j) = dl;
        double qtbpj = 0.;
        for (int k = j; k < n; k++) {
            double sk = QL(Q_SDIAG, k);
            if (sk == 0) break;
            double rkk = QR(k, k), sine, cosine;
            if (__builtin_fabs(rkk) < __builtin_fabs(sk)) {
                double cotan = rkk / sk;
                sine = 0.5 / fsq_sqrt(.25 + .25 * cotan * cotan);
                cosine = sine * cotan;
            } else {
                double tang = sk / rkk;
                cosine = 0.5 / fsq_sqrt(.25 + .25 * tang * tang);
                sine = cosine * tang;
            }
            QR(k, k) = cosine * rkk + sine * sk;
            double wk = QL(Q_TMP2, k);
            double temp = cosine * wk + sine * qtbpj;
            qtbpj = -sine * wk + cosine * qtbpj;
            QL(Q_TMP2, k) = temp;
            for (int i = k + 1; i < n; i++) {
                double rik = QR(i, k), si = QL(Q_SDIAG, i);
                double t = cosine * rik + sine * si;
                QL(Q_SDIAG, i) = -sine * rik + cosine * si;
                QR(i, k) = t;
            }
        }
        QL(Q_SDIAG, j) = QR(j, j);
        if (!ALIASED) QR(j, j) = QL(Q_TMP3, j);
    }
    int nsing = n;
    for (int j = n - 1; j >= 0; j--)
        if (QL(Q_SDIAG, j) == 0) nsing = j;
    for (int j = nsing; j < n; j++) QL(Q_TMP2, j) = 0;
    if (nsing >= 1) {
        QL(Q_TMP2, nsing - 1) = QL(Q_TMP2, nsing - 1) / QL(Q_SDIAG, nsing - 1);
        for (int j = nsing - 2; j >= 0; j--) {
            double s = 0.0;
            for (int i = j + 1; i < nsing; i++) s += QR(i, j) * QL(Q_TMP2, i);
            QL(Q_TMP2, j) = (QL(Q_TMP2, j) - s) / QL(Q_SDIAG, j);
        }
    }
    for (int j = 0; j < n; j++) QL(Q_XLM, nib_get(ipvt, j)) = QL(Q_TMP2, j);
    if (ALIASED)        // x aliases numpy.diagonal(r): r[m][m] = x[m] (mpfit.py:1915,1976-1977)
        for (int j = 0; j < n; j++) QR(j, j) = QL(Q_XLM, j);
}

// mpfit.lmpar (mpfit.py:2077-2190) on the LDS state; returns par; step left in Q_XLM
template <bool ALIASED>
FSQ_DEV double quad_lmpar(double* lds, int quad, unsigned ipvt, double delta, double par)
{
    const int n = FSQ_NP;
    double wa1[FSQ_NP], wa2[FSQ_NP];
    int nsing = n;
    double dmax = __builtin_fabs(QR(0, 0));
#pragma unroll
    for (int j = 1; j < n; j++) dmax = np_max2(dmax, __builtin_fabs(QR(j, j)));
    const double rthresh = dmax * FSQ_MACHEP;
#pragma unroll
    for (int j = n - 1; j >= 0; j--)
        if (__builtin_fabs(QR(j, j)) < rthresh) nsing = j;
#pragma unroll
    for (int j = 0; j < n; j++) wa1[j] = (j < nsing) ? QL(Q_QTF, j) : 0.0;
#pragma unroll
    for (int j = n - 1; j >= 0; j--)
        if (j < nsing) {
            wa1[j] = wa1[j] / QR(j, j);
#pragma unroll
            for (int i = 0; i < n; i++)
                if (i < j) wa1[i] = wa1[i] - QR(i, j) * wa1[j];
        }
#pragma unroll
    for (int j = 0; j < n; j++) QL(Q_XLM, nib_get(ipvt, j)) = wa1[j];
#pragma unroll
    for (int j = 0; j < n; j++) wa2[j] = QL(Q_DIAG, j) * QL(Q_XLM, j);
    double dxnorm = 0.0;
#pragma unroll
    for (int j = 0; j < n; j++) dxnorm = fsq_fma(wa2[j], wa2[j], dxnorm);
    dxnorm = fsq_sqrt(dxnorm);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) return 0.;
    // wa2 is needed permuted: stash it in LDS
#pragma unroll
    for (int j = 0; j < n; j++) QL(Q_TMP, j) = wa2[j];
    double parl = 0.;
    if (nsing >= n) {
#pragma unroll
        for (int j = 0; j < n; j++) { int l = nib_get(ipvt, j); wa1[j] = QL(Q_DIAG, l) * QL(Q_TMP, l) / dxnorm; }
        wa1[0] = wa1[0] / QR(0, 0);
#pragma unroll
        for (int j = 1; j < n; j++) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < n; i++)
                if (i < j) s += QR(i, j) * wa1[i];
            wa1[j] = (wa1[j] - s) / QR(j, j);
        }
        double temp = 0.0;
#pragma unroll
        for (int j = 0; j < n; j++) temp = fsq_fma(wa1[j], wa1[j], temp);
        temp = fsq_sqrt(temp);
        parl = ((fp / delta) / temp) / temp;
    }
#pragma unroll
    for (int j = 0; j < n; j++) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < n; i++)
            if (i <= j) s += QR(i, j) * QL(Q_QTF, i);
        wa1[j] = s / QL(Q_DIAG, nib_get(ipvt, j));
    }
    double gnorm = 0.0;
#pragma unroll
    for (int j = 0; j < n; j++) gnorm = fsq_fma(wa1[j], wa1[j], gnorm);
    gnorm = fsq_sqrt(gnorm);
    double paru = gnorm / delta;
    if (paru == 0) paru = FSQ_DWARF / np_min2(delta, 0.1);
    par = np_max2(par, parl);
    par = np_min2(par, paru);
    if (par == 0) par = gnorm / dxnorm;
    for (int iter = 1;; iter++) {
        if (par == 0) par = np_max2(FSQ_DWARF, paru * 0.001);
        double temp = fsq_sqrt(par);
#pragma unroll
        for (int j = 0; j < n; j++) QL(Q_WA1, j) = temp * QL(Q_DIAG, j);
        quad_qrsolv<ALIASED>(lds, quad, ipvt, Q_WA1);
#pragma unroll
        for (int j = 0; j < n; j++) { wa2[j] = QL(Q_DIAG, j) * QL(Q_XLM, j); QL(Q_TMP, j) = wa2[j]; }
        dxnorm = 0.0;
#pragma unroll
        for (int j = 0; j < n; j++) dxnorm = fsq_fma(wa2[j], wa2[j], dxnorm);
        dxnorm = fsq_sqrt(dxnorm);
        temp = fp;
        fp = dxnorm - delta;
        if ((__builtin_fabs(fp) <= 0.1 * delta) || ((parl == 0) && (fp <= temp) && (temp < 0)) || (iter == 10)) break;
#pragma unroll
        for (int j = 0; j < n; j++) { int l = nib_get(ipvt, j); wa1[j] = QL(Q_DIAG, l) * QL(Q_TMP, l) / dxnorm; }
#pragma unroll
        for (int j = 0; j < n - 1; j++) {
            wa1[j] = wa1[j] / QL(Q_SDIAG, j);
#pragma unroll
            for (int i = 0; i < n; i++)
                if (i > j) wa1[i] = wa1[i] - QR(i, j) * wa1[j];
        }
        wa1[n - 1] = wa1[n - 1] / QL(Q_SDIAG, n - 1);
        temp = 0.0;
#pragma unroll
        for (int j = 0; j < n; j++) temp = fsq_fma(wa1[j], wa1[j], temp);
        temp = fsq_sqrt(temp);
        double parc = ((fp / delta) / temp) / temp;
        if (fp > 0) parl = np_max2(parl, par);
        if (fp < 0) paru = np_min2(paru, par);
        par = np_max2(parl, par + parc);
    }
    return par;
}


// ---------------------------------------------------------------------------------------------------
// Register-resident lmpar / qrsolv: R (logical column order) lives in 49 VGPR pairs, every index is
// static after unrolling, so one qrsolv is ~28 rotations of pure VALU work.  The only memory it touches
// is a 14-double LDS scratch per fit (STRIDE doubles apart) used to permute 7-vectors between logical
// and parameter order.  Same operations in the same order as quad_lmpar / quad_qrsolv above.
struct QuadLm {
    double r[FSQ_NP][FSQ_NP];     // upper triangle = R, diagonal = (possibly aliased) diag, strict lower = S
    double qtf[FSQ_NP];
    double dg[FSQ_NP];            // diag by parameter index
    double dgp[FSQ_NP];           // diag[ipvt[j]]
    double sdiag[FSQ_NP];
    double xp[FSQ_NP];            // solution by parameter index
};

FSQ_DEV void quadlm_load(QuadLm& q, const double* lds, int quad, unsigned ipvt)
{
#pragma unroll
    for (int i = 0; i < FSQ_NP; i++)
#pragma unroll
        for (int k = 0; k < FSQ_NP; k++) q.r[i][k] = (k >= i) ? QR(i, k) : 0.0;
#pragma unroll
    for (int j = 0; j < FSQ_NP; j++) {
        q.qtf[j] = QL(Q_QTF, j);
        q.dg[j] = QL(Q_DIAG, j);
        q.dgp[j] = QL(Q_DIAG, nib_get(ipvt, j));
        q.sdiag[j] = QL(Q_SDIAG, j);
    }
}

// scatter a logical-order vector to parameter order through the scratch (x[ipvt[j]] = v[j])
template <int STRIDE>
FSQ_DEV void quadlm_scatter(QuadLm& q, double* scr, unsigned ipvt, const double* v)
{
#pragma unroll
    for (int j = 0; j < FSQ_NP; j++) scr[nib_get(ipvt, j) * STRIDE] = v[j];
#pragma unroll
    for (int m = 0; m < FSQ_NP; m++) q.xp[m] = scr[m * STRIDE];
}

template <bool ALIASED, int STRIDE>
FSQ_DEV void quadlm_qrsolv(QuadLm& q, double* scr, unsigned ipvt, double sqrt_par)
{
    // Fully unrolled with static register indices.  The two `break`s of the reference (mpfit.py:1921-1922,
    // 1932-1933) are lane predicates (default) or, with -DFSQ_QRSOLV_SELECTS, per-element selects that keep the 28
    // rotations in one basic block - the form that was faster at one wave per SIMD; at two it costs 3 % (a quarter
    // of the instructions of a pass were v_cndmask).
    const int n = FSQ_NP;
    double wa[FSQ_NP], xsave[FSQ_NP];
#pragma unroll
    for (int j = 0; j < n; j++)
#pragma unroll
        for (int i = j + 1; i < n; i++) q.r[i][j] = q.r[j][i];
#pragma unroll
    for (int j = 0; j < n; j++) { xsave[j] = q.r[j][j]; wa[j] = q.qtf[j]; }
#ifdef FSQ_QRSOLV_SELECTS
    bool jgo = true;                                     // false once `diag[l] == 0: break` has fired
#pragma unroll
    for (int j = 0; j < n; j++) {
        const double dl = sqrt_par * q.dgp[j];          // (temp * diag)[ipvt[j]]
        jgo = jgo && !(dl == 0);
#pragma unroll
        for (int k = j; k < n; k++) q.sdiag[k] = jgo ? ((k == j) ? dl : 0.0) : q.sdiag[k];
        double qtbpj = 0.;
        bool kgo = jgo;                                  // false once `sdiag[k] == 0: break` has fired
#pragma unroll
        for (int k = j; k < n; k++) {
            const double sk = q.sdiag[k];
            kgo = kgo && !(sk == 0);
            const double rkk = q.r[k][k];
            const bool cnd = __builtin_fabs(rkk) < __builtin_fabs(sk);
            const double num = cnd ? rkk : sk, den = cnd ? sk : rkk;
            const double t = num / den;
#ifdef FSQ_NO_ROT_SHORTCUT
            const double u = 0.5 / fsq_sqrt(.25 + .25 * t * t);
#else
            const double u = fsq_half_over_sqrt_q(.25 + .25 * t * t);          // == 0.5 / sqrt(.), see fsq_devmath.h
#endif
            const double v = u * t;
            const double cosine = cnd ? v : u, sine = cnd ? u : v;
            const double nrkk = cosine * rkk + sine * sk;
            const double temp = cosine * wa[k] + sine * qtbpj;
            const double nq = -sine * wa[k] + cosine * qtbpj;
            q.r[k][k] = kgo ? nrkk : rkk;
            qtbpj = kgo ? nq : qtbpj;
            wa[k] = kgo ? temp : wa[k];
#pragma unroll
            for (int i = k + 1; i < n; i++) {
                const double rik = q.r[i][k], si = q.sdiag[i];
                const double tt = cosine * rik + sine * si;
                const double ns = -sine * rik + cosine * si;
                q.sdiag[i] = kgo ? ns : si;
                q.r[i][k] = kgo ? tt : rik;
            }
        }
        q.sdiag[j] = jgo ? q.r[j][j] : q.sdiag[j];
        if (!ALIASED) q.r[j][j] = jgo ? xsave[j] : q.r[j][j];
    }
#else
    // The reference's two `break`s (mpfit.py:1921-1922, 1932-1933) as lane predicates: a lane that has left a loop
    // simply sits out the rest of it under the exec mask (no per-element selects); the rotation code itself stays
    // straight-line, and with two waves per SIMD the short predicated blocks cost no issue slots.
    bool jgo = true;                                     // false once `diag[l] == 0: break` has fired
#pragma unroll
    for (int j = 0; j < n; j++) {
        const double dl = sqrt_par * q.dgp[j];          // (temp * diag)[ipvt[j]]
        jgo = jgo && !(dl == 0);
        if (jgo) {
#pragma unroll
            for (int k = j; k < n; k++) q.sdiag[k] = (k == j) ? dl : 0.0;
            double qtbpj = 0.;
            bool kgo = true;                             // false once `sdiag[k] == 0: break` has fired
#pragma unroll
            for (int k = j; k < n; k++) {
                const double sk = q.sdiag[k];
                kgo = kgo && !(sk == 0);
                if (kgo) {
                    const double rkk = q.r[k][k];
                    const bool cnd = __builtin_fabs(rkk) < __builtin_fabs(sk);
                    const double num = cnd ? rkk : sk, den = cnd ? sk : rkk;
                    const double t = num / den;
#ifdef FSQ_NO_ROT_SHORTCUT
                    const double u = 0.5 / fsq_sqrt(.25 + .25 * t * t);
#else
                    const double u = fsq_half_over_sqrt_q(.25 + .25 * t * t);          // == 0.5 / sqrt(.), see fsq_devmath.h
#endif
                    const double v = u * t;
                    const double cosine = cnd ? v : u, sine = cnd ? u : v;
                    q.r[k][k] = cosine * rkk + sine * sk;
                    const double temp = cosine * wa[k] + sine * qtbpj;
                    qtbpj = -sine * wa[k] + cosine * qtbpj;
                    wa[k] = temp;
#pragma unroll
                    for (int i = k + 1; i < n; i++) {
                        const double rik = q.r[i][k], si = q.sdiag[i];
                        q.r[i][k] = cosine * rik + sine * si;
                        q.sdiag[i] = -sine * rik + cosine * si;
                    }
                }
            }
            q.sdiag[j] = q.r[j][j];
            if (!ALIASED) q.r[j][j] = xsave[j];
        }
    }
#endif
    int nsing = n;
#pragma unroll
    for (int j = n - 1; j >= 0; j--)
        if (q.sdiag[j] == 0) nsing = j;
#pragma unroll
    for (int j = 0; j < n; j++)
        if (j >= nsing) wa[j] = 0;
#pragma unroll
    for (int j = n - 1; j >= 0; j--) {
        if (j == nsing - 1) wa[j] = wa[j] / q.sdiag[j];
        else if (j < nsing - 1) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < n; i++)
                if (i > j && i < nsing) s += q.r[i][j] * wa[i];
            wa[j] = (wa[j] - s) / q.sdiag[j];
        }
    }
    quadlm_scatter<STRIDE>(q, scr, ipvt, wa);
    if (ALIASED) {
#pragma unroll
        for (int m = 0; m < n; m++) q.r[m][m] = q.xp[m];      // x IS numpy.diagonal(r)
    }
}

// returns par; the step (by parameter index) is left in q.xp; scr = 14 doubles of LDS scratch, STRIDE apart
// lmpar (mpfit.py:1970-2103) in two parts so that a caller can stop after a few Newton iterations on par and
// resume later: everything that crosses an iteration is in QuadLmparSt (plus q.r upper/diagonal and q.sdiag).
struct QuadLmparSt { double par, parl, paru, fp; int iter; bool done; };

template <bool ALIASED, int STRIDE>
FSQ_DEV void quadlm_lmpar_begin(QuadLm& q, double* scr, unsigned ipvt, double delta, double par, QuadLmparSt& st)
{
    const int n = FSQ_NP;
    double wa1[FSQ_NP];
    int nsing = n;
    double dmax = __builtin_fabs(q.r[0][0]);
#pragma unroll
    for (int j = 1; j < n; j++) dmax = np_max2(dmax, __builtin_fabs(q.r[j][j]));
    const double rthresh = dmax * FSQ_MACHEP;
#pragma unroll
    for (int j = n - 1; j >= 0; j--)
        if (__builtin_fabs(q.r[j][j]) < rthresh) nsing = j;
#pragma unroll
    for (int j = 0; j < n; j++) wa1[j] = (j < nsing) ? q.qtf[j] : 0.0;
#pragma unroll
    for (int j = n - 1; j >= 0; j--)
        if (j < nsing) {
            wa1[j] = wa1[j] / q.r[j][j];
#pragma unroll
            for (int i = 0; i < n; i++)
                if (i < j) wa1[i] = wa1[i] - q.r[i][j] * wa1[j];
        }
    quadlm_scatter<STRIDE>(q, scr, ipvt, wa1);
    double dxnorm = 0.0;
#pragma unroll
    for (int m = 0; m < n; m++) { double t = q.dg[m] * q.xp[m]; dxnorm = fsq_fma(t, t, dxnorm); }
    dxnorm = fsq_sqrt(dxnorm);
    double fp = dxnorm - delta;
    st.iter = 0;
    if (fp <= 0.1 * delta) { st.par = 0.; st.parl = 0.; st.paru = 0.; st.fp = fp; st.done = true; return; }
    double parl = 0.;
    if (nsing >= n) {
#pragma unroll
        for (int j = 0; j < n; j++) wa1[j] = q.dgp[j] * (q.dgp[j] * wa1[j]) / dxnorm;   // diag[l]*wa2[l]/dxnorm, wa2 = diag*x
        wa1[0] = wa1[0] / q.r[0][0];
#pragma unroll
        for (int j = 1; j < n; j++) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < n; i++)
                if (i < j) s += q.r[i][j] * wa1[i];
            wa1[j] = (wa1[j] - s) / q.r[j][j];
        }
        double temp = 0.0;
#pragma unroll
        for (int j = 0; j < n; j++) temp = fsq_fma(wa1[j], wa1[j], temp);
        temp = fsq_sqrt(temp);
        parl = ((fp / delta) / temp) / temp;
    }
#pragma unroll
    for (int j = 0; j < n; j++) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < n; i++)
            if (i <= j) s += q.r[i][j] * q.qtf[i];
        wa1[j] = s / q.dgp[j];
    }
    double gnorm = 0.0;
#pragma unroll
    for (int j = 0; j < n; j++) gnorm = fsq_fma(wa1[j], wa1[j], gnorm);
    gnorm = fsq_sqrt(gnorm);
    double paru = gnorm / delta;
    if (paru == 0) paru = FSQ_DWARF / np_min2(delta, 0.1);
    par = np_max2(par, parl);
    par = np_min2(par, paru);
    if (par == 0) par = gnorm / dxnorm;
    st.par = par; st.parl = parl; st.paru = paru; st.fp = fp; st.done = false;
}

// runs Newton iterations until the reference's loop would break or `iter_limit` iterations (counted from the
// start of lmpar) have been done; st.done tells which
template <bool ALIASED, int STRIDE>
FSQ_DEV void quadlm_lmpar_run(QuadLm& q, double* scr, unsigned ipvt, double delta, QuadLmparSt& st, int iter_limit)
{
    const int n = FSQ_NP;
    double* scr2 = scr + 7 * STRIDE;
    double wa1[FSQ_NP], wa2[FSQ_NP];
    double par = st.par, parl = st.parl, paru = st.paru, fp = st.fp, dxnorm;
    int iter = st.iter;
    bool done = st.done;
    while (!done && iter < iter_limit) {
        iter++;
        if (par == 0) par = np_max2(FSQ_DWARF, paru * 0.001);
        double temp = fsq_sqrt(par);
        quadlm_qrsolv<ALIASED, STRIDE>(q, scr, ipvt, temp);
#pragma unroll
        for (int m = 0; m < n; m++) wa2[m] = q.dg[m] * q.xp[m];
        dxnorm = 0.0;
#pragma unroll
        for (int m = 0; m < n; m++) dxnorm = fsq_fma(wa2[m], wa2[m], dxnorm);
        dxnorm = fsq_sqrt(dxnorm);
        temp = fp;
        fp = dxnorm - delta;
        if ((__builtin_fabs(fp) <= 0.1 * delta) || ((parl == 0) && (fp <= temp) && (temp < 0)) || (iter == 10)) { done = true; break; }
        // wa1 = diag[ipvt] * wa2[ipvt] / dxnorm : gather wa2 by logical position through the scratch
#pragma unroll
        for (int m = 0; m < n; m++) scr2[m * STRIDE] = wa2[m];
#pragma unroll
        for (int j = 0; j < n; j++) wa1[j] = q.dgp[j] * scr2[nib_get(ipvt, j) * STRIDE] / dxnorm;
#pragma unroll
        for (int j = 0; j < n - 1; j++) {
            wa1[j] = wa1[j] / q.sdiag[j];
#pragma unroll
            for (int i = 0; i < n; i++)
                if (i > j) wa1[i] = wa1[i] - q.r[i][j] * wa1[j];
        }
        wa1[n - 1] = wa1[n - 1] / q.sdiag[n - 1];
        temp = 0.0;
#pragma unroll
        for (int j = 0; j < n; j++) temp = fsq_fma(wa1[j], wa1[j], temp);
        temp = fsq_sqrt(temp);
        double parc = ((fp / delta) / temp) / temp;
        if (fp > 0) parl = np_max2(parl, par);
        if (fp < 0) paru = np_min2(paru, par);
        par = np_max2(parl, par + parc);
    }
    st.par = par; st.parl = parl; st.paru = paru; st.fp = fp; st.iter = iter; st.done = done;
}

template <bool ALIASED, int STRIDE>
FSQ_DEV double quadlm_lmpar(QuadLm& q, double* scr, unsigned ipvt, double delta, double par)
{
    QuadLmparSt st;
    quadlm_lmpar_begin<ALIASED, STRIDE>(q, scr, ipvt, delta, par, st);
    quadlm_lmpar_run<ALIASED, STRIDE>(q, scr, ipvt, delta, st, 10);
    return st.par;
}
