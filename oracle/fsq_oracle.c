/*
 * fsq_oracle.c - CPU restatement (plain C, fp64) of the reference's per-field hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product links, loads or calls this file:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / reported CPU baseline.  Parity status: PINNED - this restatement is checked
 * bit-for-bit (parameters, exit status, iteration and evaluation counts) against outputs
 * of the reference itself, run in the build container through oracle/refload.py and
 * stored in tests/golden/ (.npz; see oracle/gen_golden.py and tests/test_oracle_golden.py).
 *
 * What is restated (reference file:line given at each function):
 *   pflib._psf_candidates      pflib.py:217-258   (+ scipy.ndimage.median_filter 'reflect',
 *                                                    scipy.signal.correlate 'same', numpy mean/std)
 *   pflib._fit_2d_gaussian     pflib.py:180-214
 *   gaussfitter.gaussfit       agpy/gaussfitter.py:142-255, twodgaussian :63-140
 *   mpfit (driver, fdjac2, qrfac, qrsolv, lmpar, enorm)   agpy/mpfit/mpfit.py:600-1388,
 *                                                    1504-1612, 1748-1822, 1903-1978, 2077-2190
 *   pflib.illumina_s_n         pflib.py:261-281
 *   pflib.find_peptides        pflib.py:284-520   (R^2 filter, consolidation, re-key)
 *   phase_correlate            phase_correlate.py:11-196  (plain DFT based, see fsq_o_phase_correlate)
 *
 * The reference is Python 2 + numpy; its floating-point results depend on how numpy, its
 * OpenBLAS and glibc evaluate things.  The arithmetic model restated here is the one of the
 * reference as run in the build container (numpy 2.2.6, scipy-openblas 0.3.29 SkylakeX
 * kernels, glibc 2.35 FMA variants):
 *   - python builtin sum(): left-to-right, starting from 0
 *   - numpy.dot(v, v) (mpfit.enorm): OpenBLAS ddot - see dot_contig()/dot_strided()
 *   - numpy scalar x**2  = libm pow(x, 2.0)   (NOT x*x; differs in ~0.08% of arguments)
 *   - numpy array  x**2  = x*x
 *   - numpy.exp/cos/sin  = libm exp/cos/sin
 *   - numpy.mean/std/sum : pairwise summation (pairwise_sum())
 *   - qrsolv's solution vector aliases the diagonal of R (mpfit.py:1915,1976-1977)
 * Compile with -ffp-contract=off: every fused multiply-add below is explicit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef FSQ_TRACE
#include <stdio.h>
#endif

#include "fsq_oracle.h"
#include "fsq_refmath.h"

#define NP 7
#define NPIX 25

static const double MACHEP = 2.220446049250313e-16;       /* numpy.finfo(float64).eps  mpfit.py:2345 */
static const double DWARF = 2.2250738585072014e-308;      /* numpy.finfo(float64).tiny mpfit.py:2347 */

/* ---------------------------------------------------------------- numpy scalar helpers */
/* numpy.max([a,b]) / numpy.min([a,b]): NaN-propagating (loops: (a>=b || isnan(a)) ? a : b) */
static inline double np_max2(double a, double b) { return (a >= b || isnan(a)) ? a : b; }
static inline double np_min2(double a, double b) { return (a <= b || isnan(a)) ? a : b; }
static double np_max_arr(const double *v, int n) { double m = v[0]; for (int i = 1; i < n; i++) m = np_max2(m, v[i]); return m; }
static double np_min_arr(const double *v, int n) { double m = v[0]; for (int i = 1; i < n; i++) m = np_min2(m, v[i]); return m; }
/* numpy.clip(x, lo, hi) = min(max(x, lo), hi), NaN in x propagates */
static inline double np_clip(double x, double lo, double hi)
{
    double t = isnan(x) ? x : ((x > lo) ? x : lo);
    return isnan(t) ? t : ((t < hi) ? t : hi);
}

/* ---------------------------------------------------------------- OpenBLAS ddot model
 * numpy.dot(v.T, v) of mpfit.enorm (mpfit.py:1508) -> cblas_ddot -> OpenBLAS 0.3.29
 * kernel/x86_64/ddot.c, SkylakeX build.  Unit stride: blocks of 16 in 4 four-wide FMA
 * accumulators, folded (a0+a1)+a2)+a3 then (lo+hi) then horizontal add; tail is a scalar
 * FMA chain.  Non-unit stride: two accumulators, four products per step.  Verified
 * bit-exact against numpy.dot here for n = 1..31 (tests/golden/kat.npz enorm pins). */
static double dot_contig(const double *x, int n)
{
    double d = 0.0;
    int n1 = n & -16, i = 0;
    if (n1 >= 32) { /* not reached on this path (n <= 25); kept defensive: sequential */
        n1 = 0;
    }
    if (n1 == 16) {
        double S[4];
        for (int l = 0; l < 4; l++) {
            double a = fma(x[l], x[l], 0.0);
            double b = fma(x[4 + l], x[4 + l], 0.0);
            double c = fma(x[8 + l], x[8 + l], 0.0);
            double e = fma(x[12 + l], x[12 + l], 0.0);
            S[l] = ((a + b) + c) + e;
        }
        d = (S[0] + S[2]) + (S[1] + S[3]);
        i = 16;
    }
    for (; i < n; i++) d = fma(x[i], x[i], d);
    return d;
}

static double dot_strided(const double *x, int n, int inc)
{
    double t1 = 0.0, t2 = 0.0;
    int i = 0, n1 = n & -4;
    for (; i < n1; i += 4) {
        double a = x[(size_t)i * inc], b = x[(size_t)(i + 1) * inc];
        double c = x[(size_t)(i + 2) * inc], e = x[(size_t)(i + 3) * inc];
        t2 += fma(b, b, e * e);
        t1 += fma(a, a, c * c);
    }
    for (; i < n; i++) t1 = fma(x[(size_t)i * inc], x[(size_t)i * inc], t1);
    return t2 + t1;
}

/* mpfit.enorm (mpfit.py:1504-1509) */
#ifdef FSQ_TRACE
#include <stdio.h>
static double enorm_c(const double *x, int n) { double r = sqrt(dot_contig(x, n)); fprintf(stderr, "E %d 1 %.17g %.17g\n", n, r, x[0]); return r; }
static double enorm_s(const double *x, int n, int inc) { double r = sqrt(dot_strided(x, n, inc)); fprintf(stderr, "E %d %d %.17g %.17g\n", n, inc, r, x[0]); return r; }
#else
static inline double enorm_c(const double *x, int n) { return sqrt(dot_contig(x, n)); }
static inline double enorm_s(const double *x, int n, int inc) { return sqrt(dot_strided(x, n, inc)); }
#endif

/* ---------------------------------------------------------------- numpy pairwise sum
 * numpy/_core/src/umath/loops_utils.h.src  @TYPE@_pairwise_sum (PW_BLOCKSIZE 128), as used
 * by add.reduce on a contiguous float64 array (numpy.sum / mean / std). */
static double pairwise_sum(const double *a, size_t n)
{
    if (n < 8) {
        double res = 0.;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        size_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* numpy.sum / add.reduce over a contiguous float64 array: the ufunc machinery feeds the inner loop
 * NPY_BUFSIZE = 8192 elements at a time, each chunk summed pairwise and accumulated left to right
 * (verified against numpy 2.2.6 for 1-D and 2-D shapes, tests/test_oracle_golden.py). */
static double numpy_sum(const double *a, size_t n)
{
    double s = 0.0;
    for (size_t i = 0; i < n; i += 8192) s += pairwise_sum(a + i, (n - i) < 8192 ? (n - i) : 8192);
    return s;
}

/* ================================================================= candidates */
static inline int reflect_idx(int i, int n)
{   /* scipy.ndimage mode='reflect': (d c b a | a b c d | d c b a) */
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p; if (i < 0) i += p;
    return (i < n) ? i : p - 1 - i;
}

static int cmp_i64(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* pflib._psf_candidates (pflib.py:217-258).  cm_out (optional) receives the int64 response
 * image, thr_out the float64 threshold. Returns number of candidates or <0. */
/* pixel i of a frame stored as uint16 (wide = 0) or uint32 (wide = 1): the reference works on image.astype(np.int64)
 * (pflib.py:241, 443) whatever integer type it is handed */
static inline int64_t px_at(const void *img, int wide, size_t i)
{
    return wide ? (int64_t)((const uint32_t *)img)[i] : (int64_t)((const uint16_t *)img)[i];
}

static int candidates_any(const void *img, int wide, int H, int W, int med_size, const int64_t *K, int ksz,
                          double c_std, int32_t *hw_out, int cap, int64_t *cm_out, double *thr_out)
{
    if (ksz <= 0 || (ksz % 2) == 0) return FSQ_O_EINVAL;          /* pflib.py:236-239 */
    if (med_size <= 0 || H <= 0 || W <= 0) return FSQ_O_EINVAL;
    size_t N = (size_t)H * W;
    int64_t *mf = (int64_t *)malloc(N * sizeof(int64_t));
    int64_t *cm = cm_out ? cm_out : (int64_t *)malloc(N * sizeof(int64_t));
    double *xd = (double *)malloc(N * sizeof(double));
    int64_t *win = (int64_t *)malloc((size_t)med_size * med_size * sizeof(int64_t));
    if (!mf || !cm || !xd || !win) return FSQ_O_ENOMEM;
    /* scipy.ndimage.median_filter(image, size): footprint origin for even sizes is size//2 */
    int mo = med_size / 2, nwin = med_size * med_size, rank = nwin / 2;
    for (int h = 0; h < H; h++)
        for (int w = 0; w < W; w++) {
            int t = 0;
            for (int i = 0; i < med_size; i++)
                for (int j = 0; j < med_size; j++)
                    win[t++] = px_at(img, wide, (size_t)reflect_idx(h + i - mo, H) * W + reflect_idx(w + j - mo, W));
            qsort(win, nwin, sizeof(int64_t), cmp_i64);
            int64_t v = px_at(img, wide, (size_t)h * W + w), m = win[rank];
            mf[(size_t)h * W + w] = v - (m < v ? m : v);                /* pflib.py:243-245 */
        }
    int c = (ksz - 1) / 2;
    for (int h = 0; h < H; h++)
        for (int w = 0; w < W; w++) {
            int64_t s = 0;
            for (int i = 0; i < ksz; i++) {
                int hh = h + i - c;
                if (hh < 0 || hh >= H) continue;
                for (int j = 0; j < ksz; j++) {
                    int ww = w + j - c;
                    if (ww < 0 || ww >= W) continue;
                    s += mf[(size_t)hh * W + ww] * K[i * ksz + j];
                }
            }
            cm[(size_t)h * W + w] = s > 0 ? s : 0;                      /* pflib.py:247-248 */
        }
    /* numpy.mean(int64): float64 accumulation; exact while the running sum stays < 2^53 */
    double sum = 0.0;
    for (size_t i = 0; i < N; i++) sum += (double)cm[i];
    if (sum >= 9007199254740992.0) { free(mf); if (!cm_out) free(cm); free(xd); free(win); return FSQ_O_ERANGE; }
    double mean = sum / (double)N;
    /* numpy.std -> _var: x = arr - mean; x = x*x; pairwise sum; / N; sqrt */
    for (size_t i = 0; i < N; i++) { double d = (double)cm[i] - mean; xd[i] = d * d; }
    double var = numpy_sum(xd, N) / (double)N;
    double thr = mean + c_std * sqrt(var);                              /* pflib.py:250 */
    if (thr_out) *thr_out = thr;
    int n = 0;
    for (int h = 2; h < H - 2; h++)                                     /* pflib.py:252-257 */
        for (int w = 2; w < W - 2; w++) {
            if ((double)cm[(size_t)h * W + w] < thr) continue;
            if (n < cap) { hw_out[2 * n] = h; hw_out[2 * n + 1] = w; }
            n++;
        }
    free(mf); if (!cm_out) free(cm); free(xd); free(win);
    return n;
}

int fsq_o_candidates(const uint16_t *img, int H, int W, int med_size, const int64_t *K, int ksz,
                     double c_std, int32_t *hw_out, int cap, int64_t *cm_out, double *thr_out)
{
    return candidates_any(img, 0, H, W, med_size, K, ksz, c_std, hw_out, cap, cm_out, thr_out);
}

/* the same on uint32 pixels (values < 2^31) */
int fsq_o_candidates_u32(const uint32_t *img, int H, int W, int med_size, const int64_t *K, int ksz,
                         double c_std, int32_t *hw_out, int cap, int64_t *cm_out, double *thr_out)
{
    return candidates_any(img, 1, H, W, med_size, K, ksz, c_std, hw_out, cap, cm_out, thr_out);
}

/* ================================================================= model */
/* gaussfitter.twodgaussian (gaussfitter.py:63-140) on numpy.indices((5,5)); p = (b, a, p2, p3, wx, wy, rot deg).
 * NB center_y = p[2], center_x = p[3] (gaussfitter.py:100); x = row index, y = column index. */
static void model_eval(const double *p, double *g)
{
    double rota = FSQ_PI / 180. * p[6];                                 /* :115 */
    double c = fsq_ref_cos(rota), s = fsq_ref_sin(rota);
    double rcen_x = p[3] * c - p[2] * s;                                /* :116 */
    double rcen_y = p[3] * s + p[2] * c;                                /* :117 */
    for (int xi = 0; xi < 5; xi++)
        for (int yi = 0; yi < 5; yi++) {
            double x = (double)xi, y = (double)yi;
            double xp = x * c - y * s;                                  /* :128 */
            double yp = x * s + y * c;                                  /* :129 */
            double u = (rcen_x - xp) / p[4];
            double v = (rcen_y - yp) / p[5];
            double e = -(u * u + v * v) / 2.;
            g[xi * 5 + yi] = p[0] + p[1] * fsq_ref_exp(e);              /* :133-135 */
        }
}

/* optional debug trace of every evaluated parameter vector (tests only) */
static __thread double *g_trace = 0;
static __thread int g_trace_n = 0, g_trace_cap = 0;
void fsq_o_set_trace(double *buf, int cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
int fsq_o_get_trace_n(void) { return g_trace_n; }

/* residual closure f(p) = ravel(data - model) (gaussfitter.py:214) */
static void residual(const double *data, const double *p, double *r, int *nfev)
{
    double g[NPIX];
    if (g_trace && g_trace_n < g_trace_cap) { memcpy(g_trace + 7 * g_trace_n, p, 7 * sizeof(double)); g_trace_n++; }
    model_eval(p, g);
    for (int i = 0; i < NPIX; i++) r[i] = data[i] - g[i];
    (*nfev)++;
}

/* ================================================================= qrfac  (mpfit.py:1748-1822)
 * a is m x n row-major (stride n). */
/* test switch: take the norm re-computation branch of qrfac (mpfit.py:1817-1820) at every step - the branch is
 * practically unreachable on image data, this is how the GPU's implementation of it gets compared */
static int g_force_norm_recompute = 0;
void fsq_o_set_force_norm_recompute(int on) { g_force_norm_recompute = on; }

void fsq_o_qrfac(double *a, int m, int n, int pivot, int *ipvt, double *rdiag, double *acnorm)
{
    double wa[16];
    for (int j = 0; j < n; j++) {
        acnorm[j] = enorm_s(a + j, m, n);                               /* :1759 */
        rdiag[j] = acnorm[j]; wa[j] = rdiag[j]; ipvt[j] = j;
    }
    int minmn = m < n ? m : n;
    for (int j = 0; j < minmn; j++) {
        if (pivot) {
            double rmax = np_max_arr(rdiag + j, n - j);                 /* :1769 */
            int kmax = -1;
            for (int k = j; k < n; k++) if (rdiag[k] == rmax) { kmax = k; break; }
            if (kmax >= 0 && kmax != j) {
                int t = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = t;
                rdiag[kmax] = rdiag[j];
                wa[kmax] = wa[j];
            }
        }
        int lj = ipvt[j];
        double ajnorm = enorm_s(a + (size_t)j * n + lj, m - j, n);     /* :1789 */
        if (ajnorm == 0) break;                                         /* :1790 */
        if (a[(size_t)j * n + lj] < 0) ajnorm = -ajnorm;
        for (int i = j; i < m; i++) a[(size_t)i * n + lj] = a[(size_t)i * n + lj] / ajnorm;
        a[(size_t)j * n + lj] = a[(size_t)j * n + lj] + 1;
        for (int k = j + 1; k < n; k++) {
            int lk = ipvt[k];
            double ajj0 = a[(size_t)j * n + lj];
            if (ajj0 != 0) {                                            /* :1812 */
                double s = 0.0;
                for (int i = j; i < m; i++) s += a[(size_t)i * n + lk] * a[(size_t)i * n + lj];
                for (int i = j; i < m; i++)                             /* :1813  ajk - ajj*sum/ajj0 */
                    a[(size_t)i * n + lk] = a[(size_t)i * n + lk] - (a[(size_t)i * n + lj] * s) / ajj0;
                if (pivot && rdiag[k] != 0) {
                    double temp = a[(size_t)j * n + lk] / rdiag[k];
                    rdiag[k] = rdiag[k] * sqrt(np_max2(1. - fsq_ref_pow2(temp), 0.));   /* :1816 scalar **2 */
                    temp = rdiag[k] / wa[k];
                    if ((0.05 * temp * temp) <= MACHEP || g_force_norm_recompute) {
                        rdiag[k] = enorm_s(a + (size_t)(j + 1) * n + lk, m - j - 1, n);
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

/* ================================================================= qrsolv (mpfit.py:1903-1978)
 * r is n x n row-major.  aliased != 0 reproduces the reference: x IS the diagonal of r.
 * x_out receives the returned solution (a copy of the view's content at return time). */
static void qrsolv(double *r, int n, const int *ipvt, const double *diag, const double *qtb,
                   double *x_out, double *sdiag, int aliased)
{
    double wa[NP], xsave[NP];
    for (int j = 0; j < n; j++)                                         /* :1913-1914 */
        for (int i = j; i < n; i++) r[i * n + j] = r[j * n + i];
    for (int j = 0; j < n; j++) { xsave[j] = r[j * n + j]; wa[j] = qtb[j]; }
    for (int j = 0; j < n; j++) {
        int l = ipvt[j];
        if (diag[l] == 0) break;                                        /* :1921-1922 */
        for (int k = j; k < n; k++) sdiag[k] = 0;
        sdiag[j] = diag[l];
        double qtbpj = 0.;
        for (int k = j; k < n; k++) {
            if (sdiag[k] == 0) break;                                   /* :1932-1933 */
            double rkk = r[k * n + k], sine, cosine;
            if (fabs(rkk) < fabs(sdiag[k])) {
                double cotan = rkk / sdiag[k];
                sine = 0.5 / sqrt(.25 + .25 * cotan * cotan);
                cosine = sine * cotan;
            } else {
                double tang = sdiag[k] / rkk;
                cosine = 0.5 / sqrt(.25 + .25 * tang * tang);
                sine = cosine * tang;
            }
            r[k * n + k] = cosine * rkk + sine * sdiag[k];              /* :1945 */
            double temp = cosine * wa[k] + sine * qtbpj;
            qtbpj = -sine * wa[k] + cosine * qtbpj;
            wa[k] = temp;
            for (int i = k + 1; i < n; i++) {                           /* :1951-1954 */
                double t = cosine * r[i * n + k] + sine * sdiag[i];
                sdiag[i] = -sine * r[i * n + k] + cosine * sdiag[i];
                r[i * n + k] = t;
            }
        }
        sdiag[j] = r[j * n + j];
        if (!aliased) r[j * n + j] = xsave[j];                          /* :1956 is a no-op when aliased */
    }
    int nsing = n;
    for (int j = 0; j < n; j++) if (sdiag[j] == 0) { nsing = j; break; }
    for (int j = nsing; j < n; j++) wa[j] = 0;
    if (nsing >= 1) {
        wa[nsing - 1] = wa[nsing - 1] / sdiag[nsing - 1];
        for (int j = nsing - 2; j >= 0; j--) {
            double s = 0.0;
            for (int i = j + 1; i < nsing; i++) s += r[i * n + j] * wa[i];
            wa[j] = (wa[j] - s) / sdiag[j];
        }
    }
    for (int j = 0; j < n; j++) x_out[ipvt[j]] = wa[j];                 /* :1977 */
    if (aliased)
        for (int j = 0; j < n; j++) r[j * n + j] = x_out[j];            /* x is numpy.diagonal(r) */
}

/* ================================================================= lmpar (mpfit.py:2077-2190) */
static double lmpar(double *r, int n, const int *ipvt, const double *diag, const double *qtb,
                    double delta, double par, double *x, double *sdiag, int aliased)
{
    double wa1[NP], wa2[NP], dg[NP];
    int nsing = n;
    for (int j = 0; j < n; j++) { wa1[j] = qtb[j]; dg[j] = fabs(r[j * n + j]); }
    double rthresh = np_max_arr(dg, n) * MACHEP;                        /* :2091 */
    for (int j = 0; j < n; j++) if (dg[j] < rthresh) { nsing = j; break; }
    for (int j = nsing; j < n; j++) wa1[j] = 0;
    for (int j = nsing - 1; j >= 0; j--) {                              /* :2098-2101 */
        wa1[j] = wa1[j] / r[j * n + j];
        for (int i = 0; i < j; i++) wa1[i] = wa1[i] - r[i * n + j] * wa1[j];
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm_c(wa2, n);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) return 0.;                                   /* :2112-2113 */
    double parl = 0.;
    if (nsing >= n) {
        for (int j = 0; j < n; j++) wa1[j] = diag[ipvt[j]] * wa2[ipvt[j]] / dxnorm;
        wa1[0] = wa1[0] / r[0];
        for (int j = 1; j < n; j++) {
            double s = 0.0;
            for (int i = 0; i < j; i++) s += r[i * n + j] * wa1[i];
            wa1[j] = (wa1[j] - s) / r[j * n + j];
        }
        double temp = enorm_c(wa1, n);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; j++) {                                       /* :2131-2133 */
        double s = 0.0;
        for (int i = 0; i <= j; i++) s += r[i * n + j] * qtb[i];
        wa1[j] = s / diag[ipvt[j]];
    }
    double gnorm = enorm_c(wa1, n);
    double paru = gnorm / delta;
    if (paru == 0) paru = DWARF / np_min2(delta, 0.1);
    par = np_max2(par, parl);
    par = np_min2(par, paru);
    if (par == 0) par = gnorm / dxnorm;
    for (;;) {
        iter++;
        if (par == 0) par = np_max2(DWARF, paru * 0.001);
        double temp = sqrt(par);
        for (int j = 0; j < n; j++) wa1[j] = temp * diag[j];
        qrsolv(r, n, ipvt, wa1, qtb, x, sdiag, aliased);
        for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
        dxnorm = enorm_c(wa2, n);
        temp = fp;
        fp = dxnorm - delta;
        if ((fabs(fp) <= 0.1 * delta) || ((parl == 0) && (fp <= temp) && (temp < 0)) || (iter == 10)) break;
        for (int j = 0; j < n; j++) wa1[j] = diag[ipvt[j]] * wa2[ipvt[j]] / dxnorm;
        for (int j = 0; j < n - 1; j++) {                               /* :2170-2172 */
            wa1[j] = wa1[j] / sdiag[j];
            for (int i = j + 1; i < n; i++) wa1[i] = wa1[i] - r[i * n + j] * wa1[j];
        }
        wa1[n - 1] = wa1[n - 1] / sdiag[n - 1];
        temp = enorm_c(wa1, n);
        double parc = ((fp / delta) / temp) / temp;
        if (fp > 0) parl = np_max2(parl, par);
        if (fp < 0) paru = np_min2(paru, par);
        par = np_max2(parl, par + parc);
    }
    return par;
}

/* ================================================================= the fit
 * pflib._fit_2d_gaussian (pflib.py:180-214) -> gaussfitter.gaussfit (gaussfitter.py:142-255)
 * -> mpfit.__init__ (mpfit.py:600-1388), specialised to m=25, n=7, all parameters free. */
int fsq_o_fit_roi(const int64_t *roi, int mode, FsqOFit *out)
{
    const int n = NP, m = NPIX;
    const int aliased = (mode == FSQ_O_MODE_REF);
    double data[NPIX];
    int nfev = 0;
    /* start values & limits, pflib.py:199-213 */
    int64_t srt[NPIX], mx = roi[0], isum = 0;
    for (int i = 0; i < m; i++) { srt[i] = roi[i]; data[i] = (double)roi[i]; if (roi[i] > mx) mx = roi[i]; isum += roi[i]; }
    qsort(srt, m, sizeof(int64_t), cmp_i64);
    double mean = (double)isum / 25.0;                                  /* numpy.mean of int64 */
    double x[NP] = { (double)srt[12], (double)mx, 2.5, 2.5, 1., 1., 0. };
    const int qllim[NP] = { 1, 1, 1, 1, 1, 1, 1 };
    const int qulim[NP] = { 0, 0, 1, 1, 1, 1, 1 };
    double llim[NP] = { 0.00, ((double)mx - mean) / 3.0, 2.00, 2.00, 0.75, 0.75, 0.00 };
    double ulim[NP] = { 0.00, 0.00, 3.00, 3.00, 2.00, 2.00, 360.00 };
    for (int i = 0; i < n; i++) {                                       /* gaussfitter.py:202-204 */
        if (x[i] > ulim[i] && qulim[i]) x[i] = ulim[i];
        if (x[i] < llim[i] && qllim[i]) x[i] = llim[i];
    }
    memset(out, 0, sizeof(*out));
    /* mpfit.py:956-964 limit consistency checks (return with status 0, params = start) */
    for (int i = 0; i < n; i++)
        if ((qllim[i] && x[i] < llim[i]) || (qulim[i] && x[i] > ulim[i])) {
            for (int k = 0; k < n; k++) out->p[k] = x[k];
            out->status = 0; out->niter = 0; out->nfev = 0; out->fnorm = -1.;
            return 0;
        }
    double fvec[NPIX], fjac[NPIX * NP], wa4[NPIX];
    double diag[NP], qtf[NP], wa1[NP], wa2[NP], wa3[NP], acnorm[NP], rdiag[NP], xlm[NP], sdiag[NP];
    double R[NP * NP];
    int ipvt[NP];
    residual(data, x, fvec, &nfev);                                     /* :999 */
    double fnorm = enorm_c(fvec, m), fnorm1 = -1.;                      /* :1019 */
    double par = 0., delta = 0., xnorm = 0., gnorm = 0.;
    int niter = 1, status = 0;
    const double ftol = 1e-10, xtol = 1e-10, gtol = 1e-10, factor = 100.;
    const int maxiter = 200;
    for (int j = 0; j < n; j++) qtf[j] = 0.;

    for (;;) {                                                          /* outer loop :1030 */
        /* fdjac2 :1512-1612 */
        const double eps = 1.4901161193847656e-08;                      /* sqrt(machep) */
        double hstep[NP];
        for (int j = 0; j < n; j++) {
            hstep[j] = eps * fabs(x[j]);
            if (hstep[j] == 0) hstep[j] = eps;                          /* :1576 */
        }
        for (int j = 0; j < n; j++)
            if (qulim[j] && (x[j] > ulim[j] - hstep[j])) hstep[j] = -hstep[j];   /* :1584-1587 */
        for (int j = 0; j < n; j++) {
            double xp[NP], fp_[NPIX];
            for (int k = 0; k < n; k++) xp[k] = x[k];
            xp[j] = xp[j] + hstep[j];
            residual(data, xp, fp_, &nfev);
            for (int i = 0; i < m; i++) fjac[i * n + j] = (fp_[i] - fvec[i]) / hstep[j];   /* :1599 */
        }
        /* pegged parameters :1073-1091 */
        int lpeg[NP], upeg[NP], nlpeg = 0, nupeg = 0;
        for (int j = 0; j < n; j++) { lpeg[j] = qllim[j] && (x[j] == llim[j]); nlpeg += lpeg[j]; }
        for (int j = 0; j < n; j++) { upeg[j] = qulim[j] && (x[j] == ulim[j]); nupeg += upeg[j]; }
        for (int j = 0; j < n; j++) if (lpeg[j]) {
            double s = 0.0;
            for (int i = 0; i < m; i++) s += fvec[i] * fjac[i * n + j];
            if (s > 0) for (int i = 0; i < m; i++) fjac[i * n + j] = 0;
        }
        for (int j = 0; j < n; j++) if (upeg[j]) {
            double s = 0.0;
            for (int i = 0; i < m; i++) s += fvec[i] * fjac[i * n + j];
            if (s < 0) for (int i = 0; i < m; i++) fjac[i * n + j] = 0;
        }
        fsq_o_qrfac(fjac, m, n, 1, ipvt, rdiag, acnorm);                /* :1094 */
        if (niter == 1) {                                               /* :1099-1110 */
            for (int j = 0; j < n; j++) { diag[j] = acnorm[j]; if (diag[j] == 0) diag[j] = 1.; }
            for (int j = 0; j < n; j++) wa3[j] = diag[j] * x[j];
            xnorm = enorm_c(wa3, n);
            delta = factor * xnorm;
            if (delta == 0.) delta = factor;
        }
        /* (q transpose)*fvec :1114-1124 */
        for (int i = 0; i < m; i++) wa4[i] = fvec[i];
        for (int j = 0; j < n; j++) {
            int lj = ipvt[j];
            double temp3 = fjac[j * n + lj];
            if (temp3 != 0) {
                double s = 0.0;
                for (int i = j; i < m; i++) s += fjac[i * n + lj] * wa4[i];
                for (int i = j; i < m; i++) wa4[i] = wa4[i] - (fjac[i * n + lj] * s) / temp3;
            }
            fjac[j * n + lj] = rdiag[j];
            qtf[j] = wa4[j];
        }
        for (int i = 0; i < n; i++)                                     /* :1127-1132 */
            for (int k = 0; k < n; k++) R[i * n + k] = fjac[i * n + ipvt[k]];
        gnorm = 0.;                                                     /* :1142-1148 */
        if (fnorm != 0)
            for (int j = 0; j < n; j++) {
                int l = ipvt[j];
                if (acnorm[l] != 0) {
                    double s = 0.0;
                    for (int i = 0; i <= j; i++) s += R[i * n + j] * qtf[i];
                    s = s / fnorm;
                    gnorm = np_max2(gnorm, fabs(s / acnorm[l]));
                }
            }
        if (gnorm <= gtol) { status = 4; break; }                       /* :1151 */
        for (int j = 0; j < n; j++) diag[j] = (diag[j] > acnorm[j]) ? diag[j] : acnorm[j];   /* :1160 */

        for (;;) {                                                      /* inner loop :1163 */
#ifdef FSQ_TRACE
            double par_in = par;
#endif
            par = lmpar(R, n, ipvt, diag, qtf, delta, par, xlm, sdiag, aliased);
#ifdef FSQ_TRACE
            fprintf(stderr, "C  lmpar delta=%.17g par_in=%.17g par_out=%.17g x=[", delta, par_in, par);
            for (int j = 0; j < n; j++) fprintf(stderr, "%.17g, ", xlm[j]);
            fprintf(stderr, "]\n");
#endif
            for (int j = 0; j < n; j++) wa1[j] = -xlm[j];
            double alpha = 1.;
            if (nlpeg > 0) {                                            /* :1187-1188 */
                double mxw = np_max_arr(wa1, n);
                for (int j = 0; j < n; j++) if (lpeg[j]) wa1[j] = np_clip(wa1[j], 0., mxw);
            }
            if (nupeg > 0) {                                            /* :1189-1190 */
                double mnw = np_min_arr(wa1, n);
                for (int j = 0; j < n; j++) if (upeg[j]) wa1[j] = np_clip(wa1[j], mnw, 0.);
            }
            {                                                           /* :1192-1202 */
                int any = 0; double tmin = 0.;
                for (int j = 0; j < n; j++)
                    if ((fabs(wa1[j]) > MACHEP) && qllim[j] && ((x[j] + wa1[j]) < llim[j])) {
                        double t = (llim[j] - x[j]) / wa1[j];
                        tmin = any ? np_min2(tmin, t) : t; any = 1;
                    }
                if (any) alpha = np_min2(alpha, tmin);
                any = 0;
                for (int j = 0; j < n; j++)
                    if ((fabs(wa1[j]) > MACHEP) && qulim[j] && ((x[j] + wa1[j]) > ulim[j])) {
                        double t = (ulim[j] - x[j]) / wa1[j];
                        tmin = any ? np_min2(tmin, t) : t; any = 1;
                    }
                if (any) alpha = np_min2(alpha, tmin);
            }
            for (int j = 0; j < n; j++) { wa1[j] = wa1[j] * alpha; wa2[j] = x[j] + wa1[j]; }   /* :1215-1216 */
            for (int j = 0; j < n; j++) {                               /* :1220-1231 */
                double sgnu = (ulim[j] >= 0) * 2. - 1., sgnl = (llim[j] >= 0) * 2. - 1.;
                double ulim1 = ulim[j] * (1 - sgnu * MACHEP) - (ulim[j] == 0) * MACHEP;
                double llim1 = llim[j] * (1 + sgnl * MACHEP) + (llim[j] == 0) * MACHEP;
                if (qulim[j] && (wa2[j] >= ulim1)) wa2[j] = ulim[j];
                if (qllim[j] && (wa2[j] <= llim1)) wa2[j] = llim[j];
            }
            for (int j = 0; j < n; j++) wa3[j] = diag[j] * wa1[j];
            double pnorm = enorm_c(wa3, n);
            if (niter == 1) delta = np_min2(delta, pnorm);              /* :1237-1238 */
            residual(data, wa2, wa4, &nfev);                            /* :1245 */
            fnorm1 = enorm_c(wa4, m);
            double actred = -1.;                                        /* :1253-1255 */
            if ((0.1 * fnorm1) < fnorm) actred = -fsq_ref_pow2(fnorm1 / fnorm) + 1.;
            for (int j = 0; j < n; j++) {                               /* :1259-1261 */
                wa3[j] = 0;
                double w = wa1[ipvt[j]];
                for (int i = 0; i <= j; i++) wa3[i] = wa3[i] + R[i * n + j] * w;
            }
            double aw[NP];
            for (int j = 0; j < n; j++) aw[j] = alpha * wa3[j];
            double temp1 = enorm_c(aw, n) / fnorm;
            double temp2 = (sqrt(alpha * par) * pnorm) / fnorm;
            double prered = temp1 * temp1 + (temp2 * temp2) / 0.5;
            double dirder = -(temp1 * temp1 + temp2 * temp2);
            double ratio = 0.;
            if (prered != 0) ratio = actred / prered;
            if (ratio <= 0.25) {                                        /* :1276-1288 */
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
            if (ratio >= 0.0001) {                                      /* :1291-1298 */
                for (int j = 0; j < n; j++) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
                for (int i = 0; i < m; i++) fvec[i] = wa4[i];
                xnorm = enorm_c(wa2, n);
                fnorm = fnorm1;
                niter = niter + 1;
            }
            status = 0;                                                 /* self.status reset by call() :1245 */
            int c1 = (fabs(actred) <= ftol) && (prered <= ftol) && (0.5 * ratio <= 1);
            if (c1) status = 1;
            if (delta <= xtol * xnorm) status = 2;
            if (c1 && (status == 2)) status = 3;
            if (status != 0) break;
            if (niter >= maxiter) status = 5;                           /* :1313-1323 */
            if ((fabs(actred) <= MACHEP) && (prered <= MACHEP) && (0.5 * ratio <= 1)) status = 6;
            if (delta <= MACHEP * xnorm) status = 7;
            if (gnorm <= MACHEP) status = 8;
            if (status != 0) break;
            if (ratio >= 0.0001) break;                                 /* :1326 */
            int fin = isfinite(ratio);                                  /* :1330-1335 */
            for (int j = 0; j < n; j++) fin = fin && isfinite(wa1[j]) && isfinite(wa2[j]) && isfinite(x[j]);
            if (!fin) { status = -16; break; }
        }
        if (status != 0) break;
    }
    if (status > 0) {                                                   /* :1351-1355 */
        residual(data, x, fvec, &nfev);
        fnorm = enorm_c(fvec, m);
    }
    fnorm = np_max2(fnorm, fnorm1);                                     /* :1357-1359 */
    fnorm = fsq_ref_pow(fnorm, 2.);
    for (int k = 0; k < n; k++) out->p[k] = x[k];
    out->status = status; out->niter = niter; out->nfev = nfev; out->fnorm = fnorm;
    return 0;
}

/* fit image from the final parameters (gaussfitter.py:253) */
void fsq_o_model(const double *p, double *g25) { model_eval(p, g25); }

/* ================================================================= metrics */
/* pflib.illumina_s_n (pflib.py:261-281) on a 5x5 ROI */
double fsq_o_illumina_s_n(const int64_t *s)
{
    int64_t op[16], mx = s[0], isum = 0;
    int t = 0;
    for (int w = 0; w < 5; w++) op[t++] = s[w];
    for (int w = 0; w < 5; w++) op[t++] = s[20 + w];
    for (int h = 1; h < 4; h++) { op[t++] = s[h * 5]; op[t++] = s[h * 5 + 4]; }
    for (int i = 0; i < 25; i++) if (s[i] > mx) mx = s[i];
    for (int i = 0; i < 16; i++) isum += op[i];
    double mean = (double)isum / 16.0, xd[16];
    for (int i = 0; i < 16; i++) { double d = (double)op[i] - mean; xd[i] = d * d; }
    double sd = sqrt(numpy_sum(xd, 16) / 16.0);
    return ((double)mx - mean) / sd;
}

/* r_2, rmse, s_n and image coordinates of one fitted candidate (pflib.py:461-473) */
void fsq_o_fit_metrics(const int64_t *roi, const double *p, int h, int w, FsqORow *row)
{
    double fit[NPIX], sub[NPIX];
    model_eval(p, fit);
    int64_t isum = 0;
    for (int i = 0; i < NPIX; i++) { sub[i] = (double)roi[i]; isum += roi[i]; }
    double mean = (double)isum / 25.0;
    double num = 0.0, den = 0.0, rm = 0.0;
    for (int i = 0; i < NPIX; i++) { double d = sub[i] - fit[i]; num += d * d; }          /* array **2 */
    for (int i = 0; i < NPIX; i++) { double d = sub[i] - mean; den += d * d; }
    for (int i = 0; i < NPIX; i++) rm += fsq_ref_pow2(sub[i] - fit[i]);                   /* scalar **2 */
    row->h0 = p[2] + h - 2.5;                                           /* pflib.py:199,461 */
    row->w0 = p[3] + w - 2.5;
    row->H = p[0]; row->A = p[1]; row->sigma_h = p[4]; row->sigma_w = p[5]; row->theta = p[6];
    row->r2 = 1.0 - num / den;
    row->rmse = sqrt(rm / 25.0);
    row->s_n = fsq_o_illumina_s_n(roi);
    row->h = h; row->w = w;
}

/* ================================================================= consolidation (pflib.py:477-519)
 * rows[i] are the fitted candidates in raster order (already R^2-filtered by the caller or not:
 * rows with !(r2 < thr) are kept, so NaN passes as in the reference).  keep_idx receives the
 * indices of surviving rows in final dict order, key_hw their re-keyed (rounded) coordinates.
 * Returns number kept, or FSQ_O_EASSERT if the reference's assert at pflib.py:518 would fire. */
typedef struct { int alive; int row; } Bin;

static double py2_round(double x) { return round(x); }                 /* half away from zero */
static double py3_round(double x) { return nearbyint(x); }             /* half to even */

int fsq_o_consolidate(const FsqORow *rows, int n, int H, int W, double r2_thr, int radius, int py2,
                      int32_t *keep_idx, int32_t *key_hw)
{
    if (radius < 2) return FSQ_O_EINVAL;                                /* pflib.py:431-432 */
    int32_t *grid = (int32_t *)malloc((size_t)H * W * sizeof(int32_t));
    int32_t *order = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    if (!grid || !order) return FSQ_O_ENOMEM;
    for (size_t i = 0; i < (size_t)H * W; i++) grid[i] = -1;
    int no = 0;
    for (int i = 0; i < n; i++) {
        if (rows[i].r2 < r2_thr) continue;                              /* :466 */
        size_t g = (size_t)rows[i].h * W + rows[i].w;
        if (grid[g] < 0) { grid[g] = i; order[no++] = i; }              /* setdefault :477 */
    }
    double rr = (double)radius * radius;                                /* python int**2 */
    for (int oi = 0; oi < no; oi++) {                                   /* :479 snapshot of items() */
        int i = order[oi];
        int h = rows[i].h, w = rows[i].w;
        if (grid[(size_t)h * W + w] != i) continue;                     /* deleted */
        int h_lo = h - radius - 2 > 0 ? h - radius - 2 : 0, h_hi = h + radius + 3 < H ? h + radius + 3 : H;
        int w_lo = w - radius - 2 > 0 ? w - radius - 2 : 0, w_hi = w + radius + 3 < W ? w + radius + 3 : W;
        int dead = 0;
        for (int hd = h_lo; hd < h_hi && !dead; hd++)
            for (int wd = w_lo; wd < w_hi; wd++) {
                if (hd == h && wd == w) continue;
                int k = grid[(size_t)hd * W + wd];
                if (k < 0) continue;
                double dh = rows[i].h0 - rows[k].h0, dw = rows[i].w0 - rows[k].w0;
                if (fsq_ref_pow2(dh) + fsq_ref_pow2(dw) > rr) continue; /* numpy scalar **2 :505 */
                if (rows[i].r2 > rows[k].r2) grid[(size_t)hd * W + wd] = -1;
                else { grid[(size_t)h * W + w] = -1; dead = 1; break; }
            }
    }
    /* re-key :514-519. dict order: surviving original keys keep their insertion slot; a re-keyed
     * entry is deleted and appended at the end. */
    int nk = 0, rc = 0;
    int32_t *moved = (int32_t *)malloc((size_t)(no > 0 ? no : 1) * sizeof(int32_t));
    int nm = 0;
    for (int oi = 0; oi < no; oi++) {
        int i = order[oi];
        size_t g = (size_t)rows[i].h * W + rows[i].w;
        if (grid[g] != i) continue;
        double rh = py2 ? py2_round(rows[i].h0) : py3_round(rows[i].h0);
        double rw = py2 ? py2_round(rows[i].w0) : py3_round(rows[i].w0);
        int hr = (int)rh, wr = (int)rw;
        if (hr != rows[i].h || wr != rows[i].w) {
            grid[g] = -1;
            if (hr >= 0 && hr < H && wr >= 0 && wr < W) {
                if (grid[(size_t)hr * W + wr] >= 0) rc = FSQ_O_EASSERT; /* assert :518 */
                else grid[(size_t)hr * W + wr] = i;
            }
            moved[nm++] = i;
        }
    }
    if (rc == 0) {
        for (int oi = 0; oi < no; oi++) {
            int i = order[oi];
            size_t g = (size_t)rows[i].h * W + rows[i].w;
            if (grid[g] == i) {
                int is_moved = 0;
                /* an entry re-keyed ONTO its... cannot equal its own old key; but another moved entry
                 * may now sit at this grid cell: only count rows whose own key is unchanged */
                double rh = py2 ? py2_round(rows[i].h0) : py3_round(rows[i].h0);
                double rw = py2 ? py2_round(rows[i].w0) : py3_round(rows[i].w0);
                if ((int)rh != rows[i].h || (int)rw != rows[i].w) is_moved = 1;
                if (!is_moved) { keep_idx[nk] = i; key_hw[2 * nk] = rows[i].h; key_hw[2 * nk + 1] = rows[i].w; nk++; }
            }
        }
        for (int k = 0; k < nm; k++) {
            int i = moved[k];
            double rh = py2 ? py2_round(rows[i].h0) : py3_round(rows[i].h0);
            double rw = py2 ? py2_round(rows[i].w0) : py3_round(rows[i].w0);
            keep_idx[nk] = i; key_hw[2 * nk] = (int)rh; key_hw[2 * nk + 1] = (int)rw; nk++;
        }
    }
    free(grid); free(order); free(moved);
    return rc ? rc : nk;
}

/* ================================================================= drivers */
int fsq_o_fit_rois_u16(const uint16_t *rois, int n, int mode, int n_threads, FsqOFit *out)
{
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (int i = 0; i < n; i++) {
        int64_t roi[NPIX];
        for (int k = 0; k < NPIX; k++) roi[k] = rois[(size_t)i * NPIX + k];
        fsq_o_fit_roi(roi, mode, &out[i]);
    }
    return 0;
}

/* pflib.find_peptides (pflib.py:284-520) */
static int find_peptides_any(const void *img, int wide, int H, int W, int med_size, const int64_t *K, int ksz,
                             double c_std, double r2_thr, int radius, int mode, int n_threads,
                             FsqORow *rows_out, FsqOFit *fits_out, int32_t *keep_idx, int32_t *key_hw,
                             int cap, int32_t *n_cand, int32_t *n_keep)
{
    if (radius < 2) return FSQ_O_EINVAL;
    int32_t *hw = (int32_t *)malloc((size_t)cap * 2 * sizeof(int32_t));
    if (!hw) return FSQ_O_ENOMEM;
    int n = candidates_any(img, wide, H, W, med_size, K, ksz, c_std, hw, cap, NULL, NULL);
    if (n < 0) { free(hw); return n; }
    *n_cand = n;
    if (n > cap) { free(hw); return FSQ_O_ERANGE; }
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (int i = 0; i < n; i++) {
        int64_t roi[NPIX];
        int h = hw[2 * i], w = hw[2 * i + 1];
        for (int a = 0; a < 5; a++)
            for (int b = 0; b < 5; b++) roi[a * 5 + b] = px_at(img, wide, (size_t)(h - 2 + a) * W + (w - 2 + b));
        FsqOFit f;
        fsq_o_fit_roi(roi, mode, &f);
        if (fits_out) fits_out[i] = f;
        fsq_o_fit_metrics(roi, f.p, h, w, &rows_out[i]);
    }
    int nk = fsq_o_consolidate(rows_out, n, H, W, r2_thr, radius, 1, keep_idx, key_hw);
    free(hw);
    if (nk < 0) return nk;
    *n_keep = nk;
    return 0;
}

int fsq_o_find_peptides(const uint16_t *img, int H, int W, int med_size, const int64_t *K, int ksz,
                        double c_std, double r2_thr, int radius, int mode, int n_threads,
                        FsqORow *rows_out, FsqOFit *fits_out, int32_t *keep_idx, int32_t *key_hw,
                        int cap, int32_t *n_cand, int32_t *n_keep)
{
    return find_peptides_any(img, 0, H, W, med_size, K, ksz, c_std, r2_thr, radius, mode, n_threads, rows_out, fits_out,
                             keep_idx, key_hw, cap, n_cand, n_keep);
}

/* the same on uint32 pixels (values < 2^31) */
int fsq_o_find_peptides_u32(const uint32_t *img, int H, int W, int med_size, const int64_t *K, int ksz,
                            double c_std, double r2_thr, int radius, int mode, int n_threads,
                            FsqORow *rows_out, FsqOFit *fits_out, int32_t *keep_idx, int32_t *key_hw,
                            int cap, int32_t *n_cand, int32_t *n_keep)
{
    return find_peptides_any(img, 1, H, W, med_size, K, ksz, c_std, r2_thr, radius, mode, n_threads, rows_out, fits_out,
                             keep_idx, key_hw, cap, n_cand, n_keep);
}

/* fits of n 5 x 5 ROIs given as int64 (any pixel width) */
int fsq_o_fit_rois_i64(const int64_t *rois, int n, int mode, int n_threads, FsqOFit *out)
{
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (int i = 0; i < n; i++) fsq_o_fit_roi(rois + (size_t)i * NPIX, mode, &out[i]);
    return 0;
}

double fsq_o_enorm(const double *x, int n, int inc) { return inc == 1 ? enorm_c(x, n) : enorm_s(x, n, inc); }
double fsq_o_pairwise_sum(const double *a, long n) { return pairwise_sum(a, (size_t)n); }
double fsq_o_numpy_sum(const double *a, long n) { return numpy_sum(a, (size_t)n); }


/* ---- Spot photometry -----------------------------------------------------------------------------
 * Spot.mexican_hat_photometry_metric, flexlibrary.py:172-210: the (2*radius+1)^2 window around (h, w), clipped at the
 * image borders by image_slice (flexlibrary.py:140-146); in the coordinates of the CLIPPED slice a pixel belongs to
 * the crown when brim <= row < diameter - brim and the same for the column, to the brim otherwise;
 * photometry = sum(crown) - len(crown) * numpy.median(brim).  Pixels are integers, so everything is exact:
 * the median of an even count is the mean of the two middle values, of an empty brim nan. */
static double mexican_hat_any(const void *img, int wide, int H, int W, int h, int w, int brim, int radius)
{
    int r0 = h - radius < 0 ? 0 : h - radius, r1 = h + radius + 1 > H ? H : h + radius + 1;
    int c0 = w - radius < 0 ? 0 : w - radius, c1 = w + radius + 1 > W ? W : w + radius + 1;
    int diameter = 2 * radius + 1;
    long long crown = 0, ncrown = 0;
    int nb = 0, cap = (r1 > r0 && c1 > c0) ? (r1 - r0) * (c1 - c0) : 0;
    int64_t *b = (int64_t *)malloc((cap > 0 ? cap : 1) * sizeof(int64_t));
    for (int r = r0; r < r1; r++)
        for (int c = c0; c < c1; c++) {
            int hh = r - r0, ww = c - c0;
            int64_t p = px_at(img, wide, (size_t)r * W + c);
            if (brim <= hh && hh < diameter - brim && brim <= ww && ww < diameter - brim) { crown += p; ncrown++; }
            else b[nb++] = p;
        }
    double med;
    if (nb == 0) med = NAN;
    else {
        qsort(b, nb, sizeof(int64_t), cmp_i64);
        med = (nb & 1) ? (double)b[nb / 2] : ((double)b[nb / 2 - 1] + (double)b[nb / 2]) / 2.0;
    }
    free(b);
    return (double)crown - (double)ncrown * med;
}
double fsq_o_mexican_hat(const uint16_t *img, int H, int W, int h, int w, int brim, int radius)
{
    return mexican_hat_any(img, 0, H, W, h, w, brim, radius);
}
/* the same on uint32 pixels (values < 2^31) */
double fsq_o_mexican_hat_u32(const uint32_t *img, int H, int W, int h, int w, int brim, int radius)
{
    return mexican_hat_any(img, 1, H, W, h, w, brim, radius);
}
