/*
 * fsq_refmath.c - restatement of the elementary functions the reference reaches through numpy
 * (TEST INFRASTRUCTURE, part of the CPU oracle).
 *
 * The reference evaluates exp/sin/cos through numpy (== glibc libm once numpy's AVX-512 loops
 * are disabled) and x**2 on numpy scalars through libm pow().  The fit is chaotic
 * at the 1-ulp level, so "the same answer as the reference" needs
 * the same bits out of these functions.  This file restates the algorithms of the build
 * container's glibc 2.35 (x86-64, the FMA ifunc variants it selects) operation by operation:
 *   exp, pow : Szabolcs Nagy, ARM optimized-routines (glibc sysdeps/ieee754/dbl-64/e_exp.c, e_pow.c)
 *   sin, cos : IBM Accurate Mathematical Library (glibc sysdeps/ieee754/dbl-64/s_sin.c)
 * Every fused multiply-add is explicit (compile with -ffp-contract=off); where they sit was read
 * off the compiled library.  tests/test_refmath.py checks bit-equality with the host libm on
 * millions of arguments, so the oracle no longer depends on which libm a host has.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "fsq_refmath.h"
#include "fsq_refmath_tables.h"

#ifndef FSQ_ORACLE_LIBM

static inline uint64_t asuint64(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double asdouble(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* ---------------------------------------------------------------- exp */
/* e_exp.c specialcase(): result in the subnormal / near-overflow range */
static double exp_specialcase(double tmp, uint64_t sbits, uint64_t ki)
{
    double scale, y;
    if ((ki & 0x80000000) == 0) {
        sbits -= 1009ull << 52;                         /* k > 0: exponent of scale might have overflowed */
        scale = asdouble(sbits);
        y = 0x1p1009 * fma(scale, tmp, scale);
        return y;
    }
    sbits += 1022ull << 52;                             /* k < 0: careful in the subnormal range */
    scale = asdouble(sbits);
    y = scale + scale * tmp;
    if (y < 1.0) {
        double hi, lo;
        lo = scale - y + scale * tmp;
        hi = 1.0 + y;
        lo = 1.0 - hi + y + lo;
        y = (hi + lo) - 1.0;
        if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
}

/* exp_inline of e_pow.c / body of e_exp.c; xtail = 0 and sign_bias = 0 for plain exp */
static double exp_core(double x, double xtail, int is_pow)
{
    uint32_t abstop = (uint32_t)(asuint64(x) >> 52) & 0x7ff;
    if (abstop - 0x3c9 >= 0x3f) {
        if (abstop - 0x3c9 >= 0x80000000u) return 1.0 + x;            /* |x| < 2^-54 */
        if (abstop >= 0x409) {                                        /* |x| >= 1024 */
            if (!is_pow) {
                if (asuint64(x) == asuint64(-INFINITY)) return 0.0;
                if (abstop >= 0x7ff) return 1.0 + x;
            }
            return (asuint64(x) >> 63) ? 0x1p-767 * 0x1p-767 : 0x1p769 * 0x1p769;   /* uflow / oflow */
        }
        abstop = 0;                                                   /* large x: handled in specialcase */
    }
    double kd = fma(x, EXP_INVLN2N, EXP_SHIFT);
    uint64_t ki = asuint64(kd);
    kd -= EXP_SHIFT;
    double r = fma(kd, EXP_NEGLN2HIN, x);
    r = fma(kd, EXP_NEGLN2LON, r);
    if (is_pow) r = xtail + r;
    uint64_t idx = 2 * (ki % 128);
    uint64_t top = ki << 45;
    double tail = asdouble(EXP_TAB[idx]);
    uint64_t sbits = EXP_TAB[idx + 1] + top;
    double r2 = r * r;
    double p23 = fma(EXP_C3, r, EXP_C2);
    double p45 = fma(r, EXP_C5, EXP_C4);
    double t = r + tail;
    double tmp = fma(p23, r2, t);
    tmp = fma(r2 * r2, p45, tmp);
    if (abstop == 0) return exp_specialcase(tmp, sbits, ki);
    double scale = asdouble(sbits);
    return fma(scale, tmp, scale);
}

double fsq_ref_exp(double x) { return exp_core(x, 0.0, 0); }

/* ---------------------------------------------------------------- pow */
/* e_pow.c log_inline (FMA form): log(|x|) as hi + *tail */
static double pow_log(uint64_t ix, double *tail)
{
    uint64_t tmp = ix - 0x3fe6955500000000ull;
    int i = (int)((tmp >> 45) % 128);
    int k = (int)((int64_t)tmp >> 52);
    uint64_t iz = ix - (tmp & (0xfffull << 52));
    double z = asdouble(iz), kd = (double)k;
    double invc = POW_LOG_TAB[i][0], logc = POW_LOG_TAB[i][1], logctail = POW_LOG_TAB[i][2];
    double r = fma(z, invc, -1.0);
    double t1 = fma(kd, POW_LN2HI, logc);
    double t2 = t1 + r;
    double lo1 = fma(kd, POW_LN2LO, logctail);
    double lo2 = t1 - t2 + r;
    double ar = POW_A[0] * r;
    double ar2 = r * ar;
    double ar3 = r * ar2;
    double hi = t2 + ar2;
    double lo3 = fma(ar, r, -ar2);
    double lo4 = t2 - hi + ar2;
    double p12 = fma(POW_A[2], r, POW_A[1]);
    double p34 = fma(POW_A[4], r, POW_A[3]);
    double p56 = fma(r, POW_A[6], POW_A[5]);
    double q = fma(p56, ar2, p34);
    q = fma(ar2, q, p12);
    double lo = ((lo1 + lo2) + lo3) + lo4;
    lo = fma(ar3, q, lo);
    double y = hi + lo;
    *tail = hi - y + lo;
    return y;
}

/* pow(x, y) for the exponents the hot path uses: y must be 2.0 (numpy scalar x**2, x**2.) */
double fsq_ref_pow(double x, double y)
{
    if (y != 2.0) return pow(x, y);                   /* not reached by the hot path */
    uint64_t ix = asuint64(x);
    uint32_t topx = (uint32_t)(ix >> 52);
    if (topx - 1 >= 0x7ff - 1) {                      /* x is 0, subnormal, inf, nan or negative */
        if (2 * ix - 1 >= 2 * asuint64(INFINITY) - 1) return x * x;   /* 0, inf, nan (y = 2 > 0) */
        ix &= 0x7fffffffffffffffull;                  /* y = 2 is an even integer: sign_bias = 0 */
        topx &= 0x7ff;
        if (topx == 0) {                              /* subnormal: normalise */
            ix = asuint64(asdouble(ix) * 0x1p52);
            ix &= 0x7fffffffffffffffull;
            ix -= 52ull << 52;
        }
    }
    double lo;
    double hi = pow_log(ix, &lo);
    double ehi = y * hi;
    double elo = fma(y, lo, fma(hi, y, -ehi));
    return exp_core(ehi, elo, 1);
}

double fsq_ref_pow2(double x) { return fsq_ref_pow(x, 2.0); }

/* ---------------------------------------------------------------- sin / cos */
static inline void sc_lookup(double u, double *sn, double *ssn, double *cs, double *ccs)
{
    int k = (int)(uint32_t)asuint64(u) * 4;            /* u.i[LOW_HALF] * 4 */
    *sn = SINCOS_TAB[k]; *ssn = SINCOS_TAB[k + 1]; *cs = SINCOS_TAB[k + 2]; *ccs = SINCOS_TAB[k + 3];
}

static inline double sc_poly_c(double xx) { return fma(fma(SC_CS6, xx, SC_CS4), xx, SC_CS2); }

/* TAYLOR_SIN (s_sin.c) */
static double taylor_sin(double xx, double x, double dx)
{
    double p = fma(SC_S5, xx, SC_S4);
    p = fma(p, xx, SC_S3);
    p = fma(p, xx, SC_S2);
    p = fma(p, xx, SC_S1);
    double t = fma(fma(p, x, -(0.5 * dx)), xx, dx);
    return x + t;
}

static double do_cos(double x, double dx)
{
    if (x < 0) dx = -dx;
    double ax = fabs(x);
    double u = SC_BIG + ax;
    x = ax - (u - SC_BIG) + dx;
    double xx = x * x;
    double s = fma(x * xx, fma(SC_SN5, xx, SC_SN3), x);
    double c = xx * sc_poly_c(xx);
    double sn, ssn, cs, ccs;
    sc_lookup(u, &sn, &ssn, &cs, &ccs);
    double cor = fma(-s, ssn, ccs);
    cor = fma(-c, cs, cor);
    cor = fma(-s, sn, cor);
    return cs + cor;
}

static double do_sin(double x, double dx)
{
    double xold = x;
    if (fabs(x) < SC_TAYLOR_LIM) return taylor_sin(x * x, x, dx);
    if (x <= 0) dx = -dx;
    double ax = fabs(x);
    double u = SC_BIG + ax;
    x = ax - (u - SC_BIG);
    double xx = x * x;
    double s = x + fma(x * xx, fma(SC_SN5, xx, SC_SN3), dx);
    double c = fma(x, dx, xx * sc_poly_c(xx));
    double sn, ssn, cs, ccs;
    sc_lookup(u, &sn, &ssn, &cs, &ccs);
    double cor = fma(s, ccs, ssn);
    cor = fma(-c, sn, cor);
    cor = fma(s, cs, cor);
    return copysign(sn + cor, xold);
}

static int reduce_sincos(double x, double *a, double *da)
{
    double t = fma(x, SC_HPINV, SC_TOINT);
    double xn = t - SC_TOINT;
    int n = (int)(asuint64(t) & 3);
    double y = fma(-xn, SC_MP1, x);
    y = fma(-xn, SC_MP2, y);
    double t2 = fma(-xn, SC_PP3, y);
    double db = fma(-SC_PP3, xn, y - t2);
    double b = fma(-xn, SC_PP4, t2);
    db = db + fma(-xn, SC_PP4, t2 - b);
    *a = b; *da = db;
    return n;
}

static double do_sincos(double a, double da, int n)
{
    double r = (n & 1) ? do_cos(a, da) : do_sin(a, da);
    return (n & 2) ? -r : r;
}

double fsq_ref_sin(double x)
{
    uint32_t k = (uint32_t)(asuint64(x) >> 32) & 0x7fffffff;
    if (k < 0x3e500000) return x;
    if (k < 0x3feb6000) return do_sin(x, 0);
    if (k < 0x400368fd) {
        double t = SC_HP0 - fabs(x);
        return copysign(do_cos(t, SC_HP1), x);
    }
    if (k < 0x419921FB) {
        double a, da;
        int n = reduce_sincos(x, &a, &da);
        return do_sincos(a, da, n);
    }
    return sin(x);                                    /* |x| >= 105414350: not reached (theta <= 360 deg) */
}

double fsq_ref_cos(double x)
{
    uint32_t k = (uint32_t)(asuint64(x) >> 32) & 0x7fffffff;
    if (k < 0x3e400000) return 1.0;
    if (k < 0x3feb6000) return do_cos(x, 0);
    if (k < 0x400368fd) {
        double y = SC_HP0 - fabs(x);
        double a = y + SC_HP1;
        double da = (y - a) + SC_HP1;
        return do_sin(a, da);
    }
    if (k < 0x419921FB) {
        double a, da;
        int n = reduce_sincos(x, &a, &da);
        return do_sincos(a, da, n + 1);
    }
    return cos(x);
}

#endif /* !FSQ_ORACLE_LIBM */
