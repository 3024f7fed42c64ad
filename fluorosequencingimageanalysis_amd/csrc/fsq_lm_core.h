// fsq_lm_core.h - one bounded 7-parameter Levenberg-Marquardt PSF fit on a 5x5 ROI, fp64, one
// GPU lane per fit.
//
// Replaces, for pflib's call (reference file:line):
//   pflib._fit_2d_gaussian            pflib.py:180-214   start values, limits
//   gaussfitter.gaussfit / twodgaussian  agpy/gaussfitter.py:142-255, 63-140   model + residuals
//   mpfit.__init__ driver             agpy/mpfit/mpfit.py:600-1388
//   mpfit.fdjac2 / qrfac / qrsolv / lmpar / enorm      :1512-1612 / 1748-1822 / 1903-1978 / 2077-2190 / 1504
// specialised to m = 25 residuals, n = 7 free parameters.  The arithmetic order is the reference's
// (python sum() left-to-right, numpy.dot through OpenBLAS ddot, libm pow for scalar **2, the
// qrsolv/diag(R) aliasing of mpfit.py:1915,1976) because the fit is chaotic at the 1-ulp level.
#pragma once
#include "fsq_devmath.h"

#define FSQ_NP 7
#define FSQ_NPIX 25

#define FSQ_MACHEP 2.220446049250313e-16
#define FSQ_DWARF 2.2250738585072014e-308
#define FSQ_PI_180 (3.141592653589793 / 180.)

// numpy.max([a,b]) / numpy.min([a,b]) propagate NaN
FSQ_DEV double np_max2(double a, double b) { return (a >= b || a != a) ? a : b; }
FSQ_DEV double np_min2(double a, double b) { return (a <= b || a != a) ? a : b; }
FSQ_DEV double np_clip(double x, double lo, double hi)
{
    double t = (x != x) ? x : ((x > lo) ? x : lo);
    return (t != t) ? t : ((t < hi) ? t : hi);
}
FSQ_DEV double fsq_sqrt(double x) { return __builtin_sqrt(x); }

// numpy.dot(v, v) as OpenBLAS 0.3.29 ddot (SkylakeX kernels) evaluates it: unit stride = 4x4 FMA lanes for the
// first 16 elements folded ((a0+a1)+a2)+a3, (lo+hi), horizontal add, then a scalar FMA tail; non-unit stride = two
// accumulators fed four products per step ("arithmetic model")
FSQ_DEV double dot7(const double* x)
{
    double d = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) d = fsq_fma(x[i], x[i], d);
    return d;
}
FSQ_DEV double dot25(const double* x)
{
    double S[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        double a = x[l] * x[l], b = x[4 + l] * x[4 + l], c = x[8 + l] * x[8 + l], e = x[12 + l] * x[12 + l];
        S[l] = ((a + b) + c) + e;
    }
    double d = (S[0] + S[2]) + (S[1] + S[3]);
#pragma unroll
    for (int i = 16; i < 25; i++) d = fsq_fma(x[i], x[i], d);
    return d;
}
// strided (Jacobian column, stride FSQ_NP) starting at row j0, n rows
FSQ_DEV double dot_col(const double* a, int n)
{
    double t1 = 0.0, t2 = 0.0;
    int i = 0, n1 = n & -4;
    for (; i < n1; i += 4) {
        double x0 = a[i * FSQ_NP], x1 = a[(i + 1) * FSQ_NP], x2 = a[(i + 2) * FSQ_NP], x3 = a[(i + 3) * FSQ_NP];
        t2 += fsq_fma(x1, x1, x3 * x3);
        t1 += fsq_fma(x0, x0, x2 * x2);
    }
    for (; i < n; i++) t1 = fsq_fma(a[i * FSQ_NP], a[i * FSQ_NP], t1);
    return t2 + t1;
}

// gaussfitter.twodgaussian on numpy.indices((5,5)) - residuals r = data - model (gaussfitter.py:214)
FSQ_DEV void fsq_model(const double* p, double* g)
{
    double s, c;
    fsq_sincos(FSQ_PI_180 * p[6], &s, &c);
    double rcen_x = p[3] * c - p[2] * s;
    double rcen_y = p[3] * s + p[2] * c;
    for (int xi = 0; xi < 5; xi++)
        for (int yi = 0; yi < 5; yi++) {
            double x = (double)xi, y = (double)yi;
            double xp = x * c - y * s;
            double yp = x * s + y * c;
            double u = (rcen_x - xp) / p[4];
            double v = (rcen_y - yp) / p[5];
            double e = -(u * u + v * v) / 2.;
            g[xi * 5 + yi] = p[0] + p[1] * fsq_exp(e);
        }
}

FSQ_DEV void fsq_residual(const double* data, const double* p, double* r)
{
    double g[FSQ_NPIX];
    fsq_model(p, g);
    for (int i = 0; i < FSQ_NPIX; i++) r[i] = data[i] - g[i];
}

// mpfit.qrfac with pivoting (mpfit.py:1748-1822); a is 25x7 row-major
FSQ_DEV void fsq_qrfac(double* a, int* ipvt, double* rdiag, double* acnorm)
{
    const int m = FSQ_NPIX, n = FSQ_NP;
    double wa[FSQ_NP];
    for (int j = 0; j < n; j++) {
        acnorm[j] = fsq_sqrt(dot_col(a + j, m));
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        ipvt[j] = j;
    }
    for (int j = 0; j < n; j++) {
        double rmax = rdiag[j];
        for (int k = j + 1; k < n; k++) rmax = np_max2(rmax, rdiag[k]);
        int kmax = -1;
        for (int k = n - 1; k >= j; k--)
            if (rdiag[k] == rmax) kmax = k;
        if (kmax >= 0 && kmax != j) {
            int t = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = t;
            rdiag[kmax] = rdiag[j];
            wa[kmax] = wa[j];
        }
        int lj = ipvt[j];
        double ajnorm = fsq_sqrt(dot_col(a + j * n + lj, m - j));
        if (ajnorm == 0) break;
        if (a[j * n + lj] < 0) ajnorm = -ajnorm;
        for (int i = j; i < m; i++) a[i * n + lj] = a[i * n + lj] / ajnorm;
        a[j * n + lj] = a[j * n + lj] + 1;
        double ajj0 = a[j * n + lj];
        for (int k = j + 1; k < n; k++) {
            int lk = ipvt[k];
            if (ajj0 != 0) {
                double s = 0.0;
                for (int i = j; i < m; i++) s += a[i * n + lk] * a[i * n + lj];
                for (int i = j; i < m; i++) a[i * n + lk] = a[i * n + lk] - (a[i * n + lj] * s) / ajj0;
                if (rdiag[k] != 0) {
                    double temp = a[j * n + lk] / rdiag[k];
                    rdiag[k] = rdiag[k] * fsq_sqrt(np_max2(1. - fsq_pow2(temp), 0.));
                    temp = rdiag[k] / wa[k];
                    if ((0.05 * temp * temp) <= FSQ_MACHEP) {
                        rdiag[k] = fsq_sqrt(dot_col(a + (j + 1) * n + lk, m - j - 1));
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

// mpfit.qrsolv (mpfit.py:1903-1978).  ALIASED: the returned solution IS numpy.diagonal(r).
template <bool ALIASED>
FSQ_DEV void fsq_qrsolv(double* r, const int* ipvt, const double* diag, const double* qtb, double* x, double* sdiag)
{
    const int n = FSQ_NP;
    double wa[FSQ_NP], xsave[FSQ_NP];
    for (int j = 0; j < n; j++)
        for (int i = j; i < n; i++) r[i * n + j] = r[j * n + i];
    for (int j = 0; j < n; j++) { xsave[j] = r[j * n + j]; wa[j] = qtb[j]; }
    for (int j = 0; j < n; j++) {
        int l = ipvt[j];
        if (diag[l] == 0) break;
        for (int k = j; k < n; k++) sdiag[k] = 0;
        sdiag[j] = diag[l];
        double qtbpj = 0.;
        for (int k = j; k < n; k++) {
            if (sdiag[k] == 0) break;
            double rkk = r[k * n + k], sine, cosine;
            if (__builtin_fabs(rkk) < __builtin_fabs(sdiag[k])) {
                double cotan = rkk / sdiag[k];
                sine = 0.5 / fsq_sqrt(.25 + .25 * cotan * cotan);
                cosine = sine * cotan;
            } else {
                double tang = sdiag[k] / rkk;
                cosine = 0.5 / fsq_sqrt(.25 + .25 * tang * tang);
                sine = cosine * tang;
            }
            r[k * n + k] = cosine * rkk + sine * sdiag[k];
            double temp = cosine * wa[k] + sine * qtbpj;
            qtbpj = -sine * wa[k] + cosine * qtbpj;
            wa[k] = temp;
            for (int i = k + 1; i < n; i++) {
                double t = cosine * r[i * n + k] + sine * sdiag[i];
                sdiag[i] = -sine * r[i * n + k] + cosine * sdiag[i];
                r[i * n + k] = t;
            }
        }
        sdiag[j] = r[j * n + j];
        if (!ALIASED) r[j * n + j] = xsave[j];
    }
    int nsing = n;
    for (int j = n - 1; j >= 0; j--)
        if (sdiag[j] == 0) nsing = j;
    for (int j = nsing; j < n; j++) wa[j] = 0;
    if (nsing >= 1) {
        wa[nsing - 1] = wa[nsing - 1] / sdiag[nsing - 1];
        for (int j = nsing - 2; j >= 0; j--) {
            double s = 0.0;
            for (int i = j + 1; i < nsing; i++) s += r[i * n + j] * wa[i];
            wa[j] = (wa[j] - s) / sdiag[j];
        }
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa[j];
    if (ALIASED)
        for (int j = 0; j < n; j++) r[j * n + j] = x[j];
}

// mpfit.lmpar (mpfit.py:2077-2190); returns the new par
template <bool ALIASED>
FSQ_DEV double fsq_lmpar(double* r, const int* ipvt, const double* diag, const double* qtb, double delta,
                         double par, double* x, double* sdiag)
{
    const int n = FSQ_NP;
    double wa1[FSQ_NP], wa2[FSQ_NP];
    int nsing = n;
    double dmax = __builtin_fabs(r[0]);
    for (int j = 1; j < n; j++) dmax = np_max2(dmax, __builtin_fabs(r[j * n + j]));
    double rthresh = dmax * FSQ_MACHEP;
    for (int j = n - 1; j >= 0; j--)
        if (__builtin_fabs(r[j * n + j]) < rthresh) nsing = j;
    for (int j = 0; j < n; j++) wa1[j] = (j < nsing) ? qtb[j] : 0.0;
    for (int j = nsing - 1; j >= 0; j--) {
        wa1[j] = wa1[j] / r[j * n + j];
        for (int i = 0; i < j; i++) wa1[i] = wa1[i] - r[i * n + j] * wa1[j];
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa1[j];
    for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = fsq_sqrt(dot7(wa2));
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) return 0.;
    double parl = 0.;
    if (nsing >= n) {
        for (int j = 0; j < n; j++) wa1[j] = diag[ipvt[j]] * wa2[ipvt[j]] / dxnorm;
        wa1[0] = wa1[0] / r[0];
        for (int j = 1; j < n; j++) {
            double s = 0.0;
            for (int i = 0; i < j; i++) s += r[i * n + j] * wa1[i];
            wa1[j] = (wa1[j] - s) / r[j * n + j];
        }
        double temp = fsq_sqrt(dot7(wa1));
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; j++) {
        double s = 0.0;
        for (int i = 0; i <= j; i++) s += r[i * n + j] * qtb[i];
        wa1[j] = s / diag[ipvt[j]];
    }
    double gnorm = fsq_sqrt(dot7(wa1));
    double paru = gnorm / delta;
    if (paru == 0) paru = FSQ_DWARF / np_min2(delta, 0.1);
    par = np_max2(par, parl);
    par = np_min2(par, paru);
    if (par == 0) par = gnorm / dxnorm;
    for (int iter = 1;; iter++) {
        if (par == 0) par = np_max2(FSQ_DWARF, paru * 0.001);
        double temp = fsq_sqrt(par);
        for (int j = 0; j < n; j++) wa1[j] = temp * diag[j];
        fsq_qrsolv<ALIASED>(r, ipvt, wa1, qtb, x, sdiag);
        for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
        dxnorm = fsq_sqrt(dot7(wa2));
        temp = fp;
        fp = dxnorm - delta;
        if ((__builtin_fabs(fp) <= 0.1 * delta) || ((parl == 0) && (fp <= temp) && (temp < 0)) || (iter == 10)) break;
        for (int j = 0; j < n; j++) wa1[j] = diag[ipvt[j]] * wa2[ipvt[j]] / dxnorm;
        for (int j = 0; j < n - 1; j++) {
            wa1[j] = wa1[j] / sdiag[j];
            for (int i = j + 1; i < n; i++) wa1[i] = wa1[i] - r[i * n + j] * wa1[j];
        }
        wa1[n - 1] = wa1[n - 1] / sdiag[n - 1];
        temp = fsq_sqrt(dot7(wa1));
        double parc = ((fp / delta) / temp) / temp;
        if (fp > 0) parl = np_max2(parl, par);
        if (fp < 0) paru = np_min2(paru, par);
        par = np_max2(parl, par + parc);
    }
    return par;
}

struct FsqLmResult {
    double p[FSQ_NP];
    int status, niter, nfev;
};

// Everything one fit carries from one outer LM iteration to the next (mpfit.py:1030 loop state).
struct FsqLmState {
    double x[FSQ_NP], diag[FSQ_NP], sdiag[FSQ_NP];
    double fvec[FSQ_NPIX];
    double llim1;                       // lower limit of the amplitude, (max - mean) / 3  (pflib.py:207-209)
    double fnorm, fnorm1, par, delta, xnorm;
    int niter, nfev;
};

FSQ_DEV double fsq_llim(int j, double llim1) { return j == 0 ? 0.00 : j == 1 ? llim1 : j < 4 ? 2.00 : j < 6 ? 0.75 : 0.00; }
FSQ_DEV double fsq_ulim(int j) { return j < 2 ? 0.00 : j < 4 ? 3.00 : j < 6 ? 2.00 : 360.00; }
FSQ_DEV bool fsq_qulim(int j) { return j >= 2; }

// start of a fit: pflib._fit_2d_gaussian start values + gaussfit clipping + mpfit's first function call
FSQ_DEV void fsq_lm_init(const double* data, double vmedian, double vmax, double vmean, FsqLmState& st)
{
    const int n = FSQ_NP;
    double x0[FSQ_NP] = {vmedian, vmax, 2.5, 2.5, 1., 1., 0.};
    st.llim1 = (vmax - vmean) / 3.0;
    for (int i = 0; i < n; i++) {                       // gaussfitter.py:202-204
        double v = x0[i];
        if (v > fsq_ulim(i) && fsq_qulim(i)) v = fsq_ulim(i);
        if (v < fsq_llim(i, st.llim1)) v = fsq_llim(i, st.llim1);
        st.x[i] = v;
        st.diag[i] = 0.; st.sdiag[i] = 0.;
    }
    fsq_residual(data, st.x, st.fvec);
    st.nfev = 1; st.niter = 1;
    st.fnorm = fsq_sqrt(dot25(st.fvec)); st.fnorm1 = -1.;
    st.par = 0.; st.delta = 0.; st.xnorm = 0.;
}

// One pass of mpfit's outer loop (mpfit.py:1030-1340): Jacobian, QR, inner LM loop.  Returns the exit
// status (0 = continue with another outer iteration).
template <bool ALIASED>
FSQ_DEV int fsq_lm_outer(const double* data, FsqLmState& st)
{
    const int n = FSQ_NP, m = FSQ_NPIX;
    double* x = st.x; double* fvec = st.fvec; double* diag = st.diag; double* sdiag = st.sdiag;
    double llim[FSQ_NP], ulim[FSQ_NP]; bool qulim[FSQ_NP];
    for (int j = 0; j < n; j++) { llim[j] = fsq_llim(j, st.llim1); ulim[j] = fsq_ulim(j); qulim[j] = fsq_qulim(j); }
    double wa4[FSQ_NPIX], fjac[FSQ_NPIX * FSQ_NP];
    double qtf[FSQ_NP], wa1[FSQ_NP], wa2[FSQ_NP], wa3[FSQ_NP], acnorm[FSQ_NP], rdiag[FSQ_NP];
    double xlm[FSQ_NP], R[FSQ_NP * FSQ_NP];
    int ipvt[FSQ_NP];
    int nfev = st.nfev, niter = st.niter, status = 0;
    double fnorm = st.fnorm, fnorm1 = st.fnorm1, par = st.par, delta = st.delta, xnorm = st.xnorm, gnorm = 0.;
    const double ftol = 1e-10, xtol = 1e-10, gtol = 1e-10, factor = 100.;
    const int maxiter = 200;
    {
        // fdjac2, mpfit.py:1512-1612
        const double eps = 1.4901161193847656e-08;
        for (int j = 0; j < n; j++) {
            double h = eps * __builtin_fabs(x[j]);
            if (h == 0) h = eps;
            if (qulim[j] && (x[j] > ulim[j] - h)) h = -h;
            double xp[FSQ_NP];
            for (int k = 0; k < n; k++) xp[k] = x[k];
            xp[j] = xp[j] + h;
            fsq_residual(data, xp, wa4); nfev++;
            for (int i = 0; i < m; i++) fjac[i * n + j] = (wa4[i] - fvec[i]) / h;
        }
        // pegged parameters, mpfit.py:1073-1091
        bool lpeg[FSQ_NP], upeg[FSQ_NP];
        int nlpeg = 0, nupeg = 0;
        for (int j = 0; j < n; j++) {
            lpeg[j] = (x[j] == llim[j]); nlpeg += lpeg[j];
            upeg[j] = qulim[j] && (x[j] == ulim[j]); nupeg += upeg[j];
        }
        for (int j = 0; j < n; j++)
            if (lpeg[j]) {
                double s = 0.0;
                for (int i = 0; i < m; i++) s += fvec[i] * fjac[i * n + j];
                if (s > 0) for (int i = 0; i < m; i++) fjac[i * n + j] = 0;
            }
        for (int j = 0; j < n; j++)
            if (upeg[j]) {
                double s = 0.0;
                for (int i = 0; i < m; i++) s += fvec[i] * fjac[i * n + j];
                if (s < 0) for (int i = 0; i < m; i++) fjac[i * n + j] = 0;
            }
        fsq_qrfac(fjac, ipvt, rdiag, acnorm);
        if (niter == 1) {
            for (int j = 0; j < n; j++) { diag[j] = acnorm[j]; if (diag[j] == 0) diag[j] = 1.; }
            for (int j = 0; j < n; j++) wa3[j] = diag[j] * x[j];
            xnorm = fsq_sqrt(dot7(wa3));
            delta = factor * xnorm;
            if (delta == 0.) delta = factor;
        }
        for (int i = 0; i < m; i++) wa4[i] = fvec[i];
        for (int j = 0; j < n; j++) {                    // (q transpose)*fvec, mpfit.py:1114-1124
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
        for (int i = 0; i < n; i++)
            for (int k = 0; k < n; k++) R[i * n + k] = fjac[i * n + ipvt[k]];
        gnorm = 0.;
        if (fnorm != 0)
            for (int j = 0; j < n; j++) {
                int l = ipvt[j];
                if (acnorm[l] != 0) {
                    double s = 0.0;
                    for (int i = 0; i <= j; i++) s += R[i * n + j] * qtf[i];
                    s = s / fnorm;
                    gnorm = np_max2(gnorm, __builtin_fabs(s / acnorm[l]));
                }
            }
        if (gnorm <= gtol) { status = 4; goto done; }
        for (int j = 0; j < n; j++) diag[j] = (diag[j] > acnorm[j]) ? diag[j] : acnorm[j];

        for (;;) {                                       // inner loop, mpfit.py:1163
            par = fsq_lmpar<ALIASED>(R, ipvt, diag, qtf, delta, par, xlm, sdiag);
            for (int j = 0; j < n; j++) wa1[j] = -xlm[j];
            double alpha = 1.;
            if (nlpeg > 0) {
                double mxw = wa1[0];
                for (int j = 1; j < n; j++) mxw = np_max2(mxw, wa1[j]);
                for (int j = 0; j < n; j++) if (lpeg[j]) wa1[j] = np_clip(wa1[j], 0., mxw);
            }
            if (nupeg > 0) {
                double mnw = wa1[0];
                for (int j = 1; j < n; j++) mnw = np_min2(mnw, wa1[j]);
                for (int j = 0; j < n; j++) if (upeg[j]) wa1[j] = np_clip(wa1[j], mnw, 0.);
            }
            {
                bool any = false; double tmin = 0.;
                for (int j = 0; j < n; j++)
                    if ((__builtin_fabs(wa1[j]) > FSQ_MACHEP) && ((x[j] + wa1[j]) < llim[j])) {
                        double t = (llim[j] - x[j]) / wa1[j];
                        tmin = any ? np_min2(tmin, t) : t; any = true;
                    }
                if (any) alpha = np_min2(alpha, tmin);
                any = false;
                for (int j = 0; j < n; j++)
                    if ((__builtin_fabs(wa1[j]) > FSQ_MACHEP) && qulim[j] && ((x[j] + wa1[j]) > ulim[j])) {
                        double t = (ulim[j] - x[j]) / wa1[j];
                        tmin = any ? np_min2(tmin, t) : t; any = true;
                    }
                if (any) alpha = np_min2(alpha, tmin);
            }
            for (int j = 0; j < n; j++) { wa1[j] = wa1[j] * alpha; wa2[j] = x[j] + wa1[j]; }
            for (int j = 0; j < n; j++) {               // snap onto the limits, mpfit.py:1220-1231
                double sgnu = (ulim[j] >= 0) * 2. - 1., sgnl = (llim[j] >= 0) * 2. - 1.;
                double ulim1 = ulim[j] * (1 - sgnu * FSQ_MACHEP) - (ulim[j] == 0) * FSQ_MACHEP;
                double llim1 = llim[j] * (1 + sgnl * FSQ_MACHEP) + (llim[j] == 0) * FSQ_MACHEP;
                if (qulim[j] && (wa2[j] >= ulim1)) wa2[j] = ulim[j];
                if (wa2[j] <= llim1) wa2[j] = llim[j];
            }
            for (int j = 0; j < n; j++) wa3[j] = diag[j] * wa1[j];
            double pnorm = fsq_sqrt(dot7(wa3));
            if (niter == 1) delta = np_min2(delta, pnorm);
            fsq_residual(data, wa2, wa4); nfev++;
            fnorm1 = fsq_sqrt(dot25(wa4));
            double actred = -1.;
            if ((0.1 * fnorm1) < fnorm) actred = -fsq_pow2(fnorm1 / fnorm) + 1.;
            for (int j = 0; j < n; j++) {
                wa3[j] = 0;
                double w = wa1[ipvt[j]];
                for (int i = 0; i <= j; i++) wa3[i] = wa3[i] + R[i * n + j] * w;
            }
            for (int j = 0; j < n; j++) wa3[j] = alpha * wa3[j];
            double temp1 = fsq_sqrt(dot7(wa3)) / fnorm;
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
                for (int j = 0; j < n; j++) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
                for (int i = 0; i < m; i++) fvec[i] = wa4[i];
                xnorm = fsq_sqrt(dot7(wa2));
                fnorm = fnorm1;
                niter = niter + 1;
            }
            status = 0;
            bool c1 = (__builtin_fabs(actred) <= ftol) && (prered <= ftol) && (0.5 * ratio <= 1);
            if (c1) status = 1;
            if (delta <= xtol * xnorm) status = 2;
            if (c1 && (status == 2)) status = 3;
            if (status != 0) break;
            if (niter >= maxiter) status = 5;
            if ((__builtin_fabs(actred) <= FSQ_MACHEP) && (prered <= FSQ_MACHEP) && (0.5 * ratio <= 1)) status = 6;
            if (delta <= FSQ_MACHEP * xnorm) status = 7;
            if (gnorm <= FSQ_MACHEP) status = 8;
            if (status != 0) break;
            if (ratio >= 0.0001) break;
            bool fin = __builtin_isfinite(ratio);
            for (int j = 0; j < n; j++) fin = fin && __builtin_isfinite(wa1[j]) && __builtin_isfinite(wa2[j]) && __builtin_isfinite(x[j]);
            if (!fin) { status = -16; break; }
        }
    }
done:
    st.nfev = nfev; st.niter = niter; st.fnorm = fnorm; st.fnorm1 = fnorm1; st.par = par; st.delta = delta; st.xnorm = xnorm;
    return status;
}

// The fit, start to finish, for one lane (no work stealing): used for stand-alone ROIs.
template <bool ALIASED>
FSQ_DEV void fsq_lm_fit(const double* data, double vmedian, double vmax, double vmean, FsqLmResult* out)
{
    FsqLmState st;
    fsq_lm_init(data, vmedian, vmax, vmean, st);
    int status;
    do { status = fsq_lm_outer<ALIASED>(data, st); } while (status == 0);
    if (status > 0) st.nfev++;        // mpfit's final function call (value unused by pflib)
    for (int k = 0; k < FSQ_NP; k++) out->p[k] = st.x[k];
    out->status = status; out->niter = st.niter; out->nfev = st.nfev;
}
