// fsq_devmath.h - fp64 elementary functions for the LM fit kernel, gfx950.
//
// The reference fit (mpfit through numpy) is chaotic at the 1-ulp level: a different last bit out
// of exp() changes a third of the fits by more than 1e-4.  To return the
// reference's numbers the kernel therefore evaluates exp / sin / cos / pow(x,2) with the SAME
// algorithms, tables and fused-multiply-add placement as the libm the reference ran on (glibc 2.35
// x86-64 FMA variants: Szabolcs Nagy's exp/pow, IBM accurate sin/cos).  All of it is plain IEEE
// fp64 (v_fma_f64 / v_mul_f64 / v_add_f64), so the GPU reproduces those bits exactly.
// Compile with -ffp-contract=off: every fma below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fsq_devtables.h"

#define FSQ_DEV __device__ __forceinline__

FSQ_DEV unsigned long long fsq_bits(double x) { return (unsigned long long)__double_as_longlong(x); }
FSQ_DEV double fsq_dbl(unsigned long long u) { return __longlong_as_double((long long)u); }
FSQ_DEV double fsq_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- exp (e_exp.c) ---------------------------------------------------------------------------
__device__ __noinline__ double fsq_exp_special(double tmp, unsigned long long sbits, unsigned long long ki)
{
    double scale, y;
    if ((ki & 0x80000000ull) == 0) {
        sbits -= 1009ull << 52;
        scale = fsq_dbl(sbits);
        return 0x1p1009 * fsq_fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;
    scale = fsq_dbl(sbits);
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

template <bool IS_POW>
FSQ_DEV double fsq_exp_core(double x, double xtail)
{
    unsigned abstop = (unsigned)(fsq_bits(x) >> 52) & 0x7ff;
    if (__builtin_expect(abstop - 0x3c9u >= 0x3fu, 0)) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;
        if (abstop >= 0x409u) {
            if (!IS_POW) {
                if (fsq_bits(x) == 0xfff0000000000000ull) return 0.0;
                if (abstop >= 0x7ffu) return 1.0 + x;
            }
            return (fsq_bits(x) >> 63) ? 0.0 : __builtin_inf();   // __math_uflow / __math_oflow values
        }
        abstop = 0;
    }
    double kd = fsq_fma(x, EXP_INVLN2N, EXP_SHIFT);
    unsigned long long ki = fsq_bits(kd);
    kd -= EXP_SHIFT;
    double r = fsq_fma(kd, EXP_NEGLN2HIN, x);
    r = fsq_fma(kd, EXP_NEGLN2LON, r);
    if (IS_POW) r = xtail + r;
    unsigned idx = 2u * ((unsigned)ki & 127u);
    unsigned long long top = ki << 45;
    double tail = fsq_dbl(FSQ_EXP_TAB[idx]);
    unsigned long long sbits = FSQ_EXP_TAB[idx + 1] + top;
    double r2 = r * r;
    double p23 = fsq_fma(EXP_C3, r, EXP_C2);
    double p45 = fsq_fma(r, EXP_C5, EXP_C4);
    double t = r + tail;
    double tmp = fsq_fma(p23, r2, t);
    tmp = fsq_fma(r2 * r2, p45, tmp);
    if (__builtin_expect(abstop == 0, 0)) return fsq_exp_special(tmp, sbits, ki);
    double scale = fsq_dbl(sbits);
    return fsq_fma(scale, tmp, scale);
}

FSQ_DEV double fsq_exp(double x) { return fsq_exp_core<false>(x, 0.0); }

// Branch-free exp for |x| < 512: fsq_exp_core's main path and nothing else, so that the pixels of a model evaluation
// form ONE basic block (independent dependency chains overlap in the VALU pipeline).  The main path is also what e_exp.c
// computes for tiny arguments (|x| < 2^-54, incl. +-0: there it returns 1 + x, and scale + scale * tmp with scale = 1,
// |tmp| <= |x| rounds to the same 1); 512 <= |x| would need the subnormal / overflow fix-ups of specialcase() or the
// inf / nan returns.  The LM model never gets there (sigma >= 0.75 and centres in [2, 3] bound the exponent by 64), so
// anything outside |x| < 512 (NaN included) only raises *bad and the caller redoes the fit with fsq_exp.
// fsq_selftest_exp compares the two over the whole double range.
FSQ_DEV double fsq_exp_bf(double x, bool* bad)
{
    *bad = *bad || !(__builtin_fabs(x) < 512.0);
    double kd = fsq_fma(x, EXP_INVLN2N, EXP_SHIFT);
    unsigned long long ki = fsq_bits(kd);
    kd -= EXP_SHIFT;
    double r = fsq_fma(kd, EXP_NEGLN2HIN, x);
    r = fsq_fma(kd, EXP_NEGLN2LON, r);
    unsigned idx = 2u * ((unsigned)ki & 127u);
    unsigned long long top = ki << 45;
    double tail = fsq_dbl(FSQ_EXP_TAB[idx]);
    unsigned long long sbits = FSQ_EXP_TAB[idx + 1] + top;
    double r2 = r * r;
    double p23 = fsq_fma(EXP_C3, r, EXP_C2);
    double p45 = fsq_fma(r, EXP_C5, EXP_C4);
    double t = r + tail;
    double tmp = fsq_fma(p23, r2, t);
    tmp = fsq_fma(r2 * r2, p45, tmp);
    double scale = fsq_dbl(sbits);
    return fsq_fma(scale, tmp, scale);
}

// ---- division by a shared divisor --------------------------------------------------------------------------
// The compiler expands every fp64 `n / d` into v_div_scale x2, v_rcp, 4 Newton fmas, mul, fma, v_div_fmas,
// v_div_fixup (13 instructions, one quarter-rate).  When v_div_scale does not rescale (VCC = 0, operands passed
// through) that sequence is rcp + 4 fma on d alone, then mul / fma / fma on n, then v_div_fixup: so for a divisor
// shared by many numerators the d-only part is hoisted (fsq_divisor) and every quotient costs 4 instructions
// (fsq_div_by) and is bit-identical to `n / d`.  v_div_scale rescales only when d is denormal or huge, the quotient
// is denormal, exponent(n) - exponent(d) >= 768, or n is tiny (biased exponent <= 53); zeros / inf / nan never
// reach the arithmetic result because v_div_fixup overrides it from (d, n) alone.  Callers keep d within
// 2^+-FSQ_DIV_ED and non-zero finite n within 2^+-FSQ_DIV_EN (exponents tracked with fsq_expo, zero/inf/nan -> 0)
// and send a fit whose values leave those ranges through the plain `/` build of the kernel instead.
#define FSQ_DIV_ED 250
#define FSQ_DIV_EN 500
struct FsqDivisor { double d, r; };
FSQ_DEV int fsq_expo(double v) { return __builtin_amdgcn_frexp_exp(v); }
FSQ_DEV FsqDivisor fsq_divisor(double d)
{
    FsqDivisor k;
    k.d = d;
    double r = __builtin_amdgcn_rcp(d);
    double e = fsq_fma(-d, r, 1.0);
    r = fsq_fma(r, e, r);
    e = fsq_fma(-d, r, 1.0);
    k.r = fsq_fma(r, e, r);
    return k;
}
FSQ_DEV bool fsq_divisor_in_range(double d) { return (unsigned)(fsq_expo(d) + FSQ_DIV_ED) <= 2u * FSQ_DIV_ED; }
FSQ_DEV double fsq_div_by(double n, const FsqDivisor& k)
{
    double q = n * k.r;
    double rem = fsq_fma(-k.d, q, n);
    q = fsq_fma(rem, k.r, q);
    return __builtin_amdgcn_div_fixup(q, k.d, n);
}
// FAST = false: the plain division (same call sites, used by the exact build of a kernel)
template <bool FAST> FSQ_DEV double fsq_div_sel(double n, const FsqDivisor& k) { return FAST ? fsq_div_by(n, k) : n / k.d; }

// ---- 0.5 / sqrt(x) for the Givens rotations of qrsolv ------------------------------------------------------------
// There x = .25 + .25 t^2 with |t| <= 1 (t = smaller / larger of two magnitudes), so x is in [0.25, 0.5] or NaN:
// the compiler's sqrt (v_rsq + two Newton steps, wrapped in a subnormal pre-scale and a 0/inf pass-through) and its
// division (v_div_scale x2 ... v_div_fixup) reduce to their cores, executed here as written - same instructions on
// the same values, hence the same bits, at 18 instead of 31 instructions.  fsq_selftest_rotation compares the two.
FSQ_DEV double fsq_half_over_sqrt_q(double x)
{
    // sqrt(x): g ~ sqrt(x), h ~ 1 / (2 sqrt(x))
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    double e = fsq_fma(-h, g, 0.5);
    g = fsq_fma(g, e, g);
    double d = fsq_fma(-g, g, x);
    h = fsq_fma(h, e, h);
    g = fsq_fma(d, h, g);
    d = fsq_fma(-g, g, x);
    g = fsq_fma(d, h, g);
    // 0.5 / g, g in [0.5, 0.7072]
    double r = __builtin_amdgcn_rcp(g);
    e = fsq_fma(-g, r, 1.0);
    r = fsq_fma(r, e, r);
    e = fsq_fma(-g, r, 1.0);
    r = fsq_fma(r, e, r);
    double q = 0.5 * r;
    const double rem = fsq_fma(-g, q, 0.5);
    return fsq_fma(rem, r, q);
}

// ---- pow(x, 2.0) (e_pow.c): what numpy computes for a float64 SCALAR ** 2 ---------------------
__device__ __noinline__ double fsq_pow2(double x)
{
    unsigned long long ix = fsq_bits(x);
    unsigned topx = (unsigned)(ix >> 52);
    if (topx - 1u >= 0x7ffu - 1u) {
        if (2 * ix - 1 >= 2 * 0x7ff0000000000000ull - 1) return x * x;   // 0, inf, nan
        ix &= 0x7fffffffffffffffull;                                     // y = 2: even, sign dropped
        topx &= 0x7ff;
        if (topx == 0) {
            ix = fsq_bits(fsq_dbl(ix) * 0x1p52);
            ix &= 0x7fffffffffffffffull;
            ix -= 52ull << 52;
        }
    }
    unsigned long long tmp = ix - 0x3fe6955500000000ull;
    int i = (int)((tmp >> 45) & 127);
    int k = (int)((long long)tmp >> 52);
    unsigned long long iz = ix - (tmp & (0xfffull << 52));
    double z = fsq_dbl(iz), kd = (double)k;
    double invc = FSQ_POW_LOG_TAB[i][0], logc = FSQ_POW_LOG_TAB[i][1], logctail = FSQ_POW_LOG_TAB[i][2];
    double r = fsq_fma(z, invc, -1.0);
    double t1 = fsq_fma(kd, POW_LN2HI, logc);
    double t2 = t1 + r;
    double lo1 = fsq_fma(kd, POW_LN2LO, logctail);
    double lo2 = t1 - t2 + r;
    double ar = POW_A[0] * r;
    double ar2 = r * ar;
    double ar3 = r * ar2;
    double hi = t2 + ar2;
    double lo3 = fsq_fma(ar, r, -ar2);
    double lo4 = t2 - hi + ar2;
    double p12 = fsq_fma(POW_A[2], r, POW_A[1]);
    double p34 = fsq_fma(POW_A[4], r, POW_A[3]);
    double p56 = fsq_fma(r, POW_A[6], POW_A[5]);
    double q = fsq_fma(p56, ar2, p34);
    q = fsq_fma(ar2, q, p12);
    double lo = ((lo1 + lo2) + lo3) + lo4;
    lo = fsq_fma(ar3, q, lo);
    double y = hi + lo;
    double tail = hi - y + lo;
    double ehi = 2.0 * y;
    double elo = fsq_fma(2.0, tail, fsq_fma(y, 2.0, -ehi));
    return fsq_exp_core<true>(ehi, elo);
}

// ---- sin / cos (s_sin.c) -----------------------------------------------------------------------
FSQ_DEV double fsq_sc_polyc(double xx) { return fsq_fma(fsq_fma(SC_CS6, xx, SC_CS4), xx, SC_CS2); }

FSQ_DEV double fsq_taylor_sin(double xx, double x, double dx)
{
    double p = fsq_fma(SC_S5, xx, SC_S4);
    p = fsq_fma(p, xx, SC_S3);
    p = fsq_fma(p, xx, SC_S2);
    p = fsq_fma(p, xx, SC_S1);
    double t = fsq_fma(fsq_fma(p, x, -(0.5 * dx)), xx, dx);
    return x + t;
}

// Index into the sin / cos table (440 doubles, 4 per entry) from the low word of big + |x|.  For every argument of the
// documented range the value is 0 .. 436 as it stands; the clamp is for lanes that run the function on GARBAGE - the step
// round executes its trial evaluation on the lanes of a tile whose queue slots are dead (reserved by the Jacobian round and not
// needed: only their tag is ever written), with whatever bits the workspace memory held; an unclamped index then read up to
// 32 GB past the table (a GPU memory fault, found by round 3's fuzz once the workspace was recycled memory instead of fresh
// zero pages).
FSQ_DEV int fsq_sincos_index(double u)
{
    const unsigned k = (unsigned)fsq_bits(u) * 4u;
    return (int)(k < 436u ? k : 436u);
}

FSQ_DEV double fsq_do_cos(double x, double dx)
{
    if (x < 0) dx = -dx;
    double ax = __builtin_fabs(x);
    double u = SC_BIG + ax;
    x = ax - (u - SC_BIG) + dx;
    double xx = x * x;
    double s = fsq_fma(x * xx, fsq_fma(SC_SN5, xx, SC_SN3), x);
    double c = xx * fsq_sc_polyc(xx);
    const int k = fsq_sincos_index(u);
    double sn = FSQ_SINCOS_TAB[k], ssn = FSQ_SINCOS_TAB[k + 1], cs = FSQ_SINCOS_TAB[k + 2], ccs = FSQ_SINCOS_TAB[k + 3];
    double cor = fsq_fma(-s, ssn, ccs);
    cor = fsq_fma(-c, cs, cor);
    cor = fsq_fma(-s, sn, cor);
    return cs + cor;
}

FSQ_DEV double fsq_do_sin(double x, double dx)
{
    double xold = x;
    if (__builtin_fabs(x) < SC_TAYLOR_LIM) return fsq_taylor_sin(x * x, x, dx);
    if (x <= 0) dx = -dx;
    double ax = __builtin_fabs(x);
    double u = SC_BIG + ax;
    x = ax - (u - SC_BIG);
    double xx = x * x;
    double s = x + fsq_fma(x * xx, fsq_fma(SC_SN5, xx, SC_SN3), dx);
    double c = fsq_fma(x, dx, xx * fsq_sc_polyc(xx));
    const int k = fsq_sincos_index(u);
    double sn = FSQ_SINCOS_TAB[k], ssn = FSQ_SINCOS_TAB[k + 1], cs = FSQ_SINCOS_TAB[k + 2], ccs = FSQ_SINCOS_TAB[k + 3];
    double cor = fsq_fma(s, ccs, ssn);
    cor = fsq_fma(-c, sn, cor);
    cor = fsq_fma(s, cs, cor);
    return __builtin_copysign(sn + cor, xold);
}

FSQ_DEV int fsq_reduce_sincos(double x, double* a, double* da)
{
    double t = fsq_fma(x, SC_HPINV, SC_TOINT);
    double xn = t - SC_TOINT;
    int n = (int)(fsq_bits(t) & 3);
    double y = fsq_fma(-xn, SC_MP1, x);
    y = fsq_fma(-xn, SC_MP2, y);
    double t2 = fsq_fma(-xn, SC_PP3, y);
    double db = fsq_fma(-SC_PP3, xn, y - t2);
    double b = fsq_fma(-xn, SC_PP4, t2);
    db = db + fsq_fma(-xn, SC_PP4, t2 - b);
    *a = b;
    *da = db;
    return n;
}

// sin and cos of the rotation angle (0 <= x < 105414350; the fit keeps theta in [0, 360] degrees)
// (one out-of-line copy per kernel; the results come back in registers - pointer outputs would put them on the stack)
struct FsqSinCos { double s, c; };
#ifndef FSQ_SINCOS_INLINE
#define FSQ_SINCOS_INLINE 0
#endif
#if FSQ_SINCOS_INLINE
__device__ __forceinline__
#else
__device__ __noinline__
#endif
FsqSinCos fsq_sincos_rv(double x)
{
    unsigned k = (unsigned)(fsq_bits(x) >> 32) & 0x7fffffffu;
    double s, c;
    if (k < 0x3feb6000u) {
        s = (k < 0x3e500000u) ? x : fsq_do_sin(x, 0.0);
        c = (k < 0x3e400000u) ? 1.0 : fsq_do_cos(x, 0.0);
    } else if (k < 0x400368fdu) {
        double y = SC_HP0 - __builtin_fabs(x);
        s = __builtin_copysign(fsq_do_cos(y, SC_HP1), x);
        double a = y + SC_HP1;
        double da = (y - a) + SC_HP1;
        c = fsq_do_sin(a, da);
    } else {
        double a, da;
        int n = fsq_reduce_sincos(x, &a, &da);
        double vs = fsq_do_sin(a, da), vc = fsq_do_cos(a, da);
        s = (n & 1) ? vc : vs;
        if (n & 2) s = -s;
        int m = n + 1;
        c = (m & 1) ? vc : vs;
        if (m & 2) c = -c;
    }
    FsqSinCos r;
    r.s = s; r.c = c;
    return r;
}
__device__ __forceinline__ void fsq_sincos(double x, double* sn_out, double* cs_out)
{
    const FsqSinCos r = fsq_sincos_rv(x);
    *sn_out = r.s;
    *cs_out = r.c;
}
