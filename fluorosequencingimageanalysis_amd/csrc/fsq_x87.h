// fsq_x87.h - the 2-vector Euclidean norm as OpenBLAS' x86-64 dnrm2 kernel evaluates it (x87 extended precision),
// in integer arithmetic, for the tracking kernel (fsq_track.hip).
//
// The reference sorts candidate ancestor/descendant pairs by scipy.spatial.distance.euclidean (flexlibrary.py:927-935),
// which is scipy.linalg.norm -> BLAS dnrm2.  The OpenBLAS kernel (kernel/x86_64/nrm2.S) squares and accumulates in
// x87 registers - every operation rounded to a 64-bit significand - takes fsqrt and stores the result as a double:
//     d = RN53( RN64( sqrt( RN64( RN64(dh*dh) + RN64(dw*dw) ) ) ) )
// In 16 % of the displacement vectors on a 1/20-pixel grid that is not the double sqrt(dh*dh + dw*dw), and the order of
// near-equal distances decides which spots are linked.  A GPU has no 80-bit type, so the chain is restated on
// (64-bit significand, exponent) pairs with 128-bit integers; plain C++ so that the host can check it against
// `long double` (tests/test_tracking.py builds a small checker from this header).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define FSQ_X87_FN __host__ __device__ inline
#else
#define FSQ_X87_FN inline
#endif

typedef unsigned __int128 fsq_u128;
struct FsqExt { uint64_t m; int e; };       // value = m * 2^e, m == 0 or bit 63 of m set

FSQ_X87_FN int fsq_clz64(uint64_t x)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

// RN64(x * x) for a finite double x
FSQ_X87_FN FsqExt fsq_x87_square(double x)
{
    union { double d; uint64_t u; } c; c.d = x;
    const uint64_t bits = c.u & 0x7fffffffffffffffull;
    FsqExt r = {0, 0};
    if (bits == 0) return r;
    int ex = (int)(bits >> 52);
    uint64_t m = bits & 0xfffffffffffffull;
    if (ex == 0) { ex = 1; } else m |= 1ull << 52;
    int e = ex - 1075;                                  // x = m * 2^e
    const int lz = fsq_clz64(m);                        // normalise (subnormals) so that bit 63 is the top bit
    m <<= lz; e -= lz;
    const fsq_u128 p = (fsq_u128)m * m;                 // 127 or 128 bits
    int sh = 64;
    uint64_t q;
    if ((uint64_t)(p >> 127)) q = (uint64_t)(p >> 64); else { q = (uint64_t)(p >> 63); sh = 63; }
    const fsq_u128 rem = p & ((((fsq_u128)1) << sh) - 1), half = ((fsq_u128)1) << (sh - 1);
    int up = rem > half || (rem == half && (q & 1));
    r.e = 2 * e + sh;
    if (up) { q++; if (q == 0) { q = 1ull << 63; r.e++; } }
    r.m = q;
    return r;
}

// RN64(a + b), a, b >= 0
FSQ_X87_FN FsqExt fsq_x87_add(FsqExt a, FsqExt b)
{
    if (a.m == 0) return b;
    if (b.m == 0) return a;
    if (b.e > a.e) { FsqExt t = a; a = b; b = t; }      // (both normalised: the larger exponent is the larger number)
    const int d = a.e - b.e;
    const fsq_u128 big = (fsq_u128)a.m << 62;
    fsq_u128 small = 0; int sticky = 0;
    if (d < 126) {
        const fsq_u128 bs = (fsq_u128)b.m << 62;
        small = bs >> d;
        sticky = (small << d) != bs;
    } else sticky = 1;
    const fsq_u128 s = big + small;                     // 126 or 127 bits
    int sh; uint64_t q;
    if ((uint64_t)(s >> 126)) { sh = 63; } else { sh = 62; }
    q = (uint64_t)(s >> sh);
    const fsq_u128 rem = s & ((((fsq_u128)1) << sh) - 1), half = ((fsq_u128)1) << (sh - 1);
    const int up = rem > half || (rem == half && (sticky || (q & 1)));
    FsqExt r; r.e = a.e - 62 + sh;
    if (up) { q++; if (q == 0) { q = 1ull << 63; r.e++; } }
    r.m = q;
    return r;
}

// RN64(sqrt(a))
FSQ_X87_FN FsqExt fsq_x87_sqrt(FsqExt a)
{
    FsqExt r = {0, 0};
    if (a.m == 0) return r;
    // radicand as an exact 127/128-bit integer N * 2^E2 with E2 even
    const int odd = a.e & 1;
    const fsq_u128 N = (fsq_u128)a.m << (odd ? 63 : 64);
    const int E2 = a.e - (odd ? 63 : 64);
    // R = floor(sqrt(N)): double estimate, one correction step in doubles, then exact fix-up
    const double nd = (double)(uint64_t)(N >> 64) * 18446744073709551616.0 + (double)(uint64_t)N;
    double rd = __builtin_sqrt(nd);
    if (rd >= 18446744073709549568.0) rd = 18446744073709549568.0;       // largest double below 2^64
    uint64_t R = (uint64_t)rd;
    {
        const fsq_u128 sq = (fsq_u128)R * R;
        const int neg = sq > N;
        const fsq_u128 diff = neg ? sq - N : N - sq;
        const double dd = (double)(uint64_t)(diff >> 64) * 18446744073709551616.0 + (double)(uint64_t)diff;
        const double corr = dd / (2.0 * (double)R);
        const uint64_t c = (uint64_t)corr;
        if (neg) R -= c; else { const uint64_t room = ~R; R += (c > room ? room : c); }
    }
    for (int it = 0; it < 8 && (fsq_u128)R * R > N; it++) R--;
    for (int it = 0; it < 8; it++) {
        if (R == ~0ull) break;
        const fsq_u128 nx = (fsq_u128)(R + 1) * (R + 1);
        if (nx > N) break;
        R++;
    }
    const fsq_u128 rem = N - (fsq_u128)R * R;           // sqrt(N) >= R + 1/2  <=>  rem > R   (a tie cannot happen)
    r.e = E2 / 2;
    if (rem > (fsq_u128)R) { R++; if (R == 0) { R = 1ull << 63; r.e++; } }
    r.m = R;
    return r;
}

// RN53(a) as a double
FSQ_X87_FN double fsq_x87_to_double(FsqExt a)
{
    if (a.m == 0) return 0.0;
    uint64_t q = a.m >> 11;
    const uint64_t rem = a.m & 2047u;
    int e = a.e + 11;
    if (rem > 1024u || (rem == 1024u && (q & 1))) { q++; if (q == (1ull << 53)) { q >>= 1; e++; } }
    // q in [2^52, 2^53): assemble the double (the distances of this path are far from the exponent limits)
    union { double d; uint64_t u; } c;
    c.u = ((uint64_t)(e + 52 + 1023) << 52) | (q & 0xfffffffffffffull);
    return c.d;
}

FSQ_X87_FN double fsq_dnrm2_2(double dh, double dw)
{
    return fsq_x87_to_double(fsq_x87_sqrt(fsq_x87_add(fsq_x87_square(dh), fsq_x87_square(dw))));
}
