// fsq_fit_rounds.hip - K3/K4 production path: the LM fit as a WAVEFRONT of rounds over all candidates.
//
// Why: one LM solve is a chain of very different pieces - the Jacobian + pivoted QR (column-parallel,
// regular) and mpfit's inner loop (lmpar with 0..10 qrsolv passes + a trial evaluation, scalar and wildly
// irregular: a quarter of the lmpar calls run all 10 passes, most run 0-1, steps get rejected and retried).
// Run in lockstep inside one wave the irregular part leaves 70% of the lanes idle (measured).  So every
// candidate keeps its solver state in HBM (queue records in structure-of-arrays form, see below) and the fit
// advances in rounds:
//   kA  "Jacobian round"  quad-cooperative (4 lanes per fit, columns in registers, see fsq_lm_quad.h):
//                         fdjac2 + qrfac + Q^T f + gradient test            mpfit.py:1064-1160
//   kB  "step round"      one lane per fit, R in registers: lmpar, bounded step, trial evaluation,
//                         trust-region update, convergence tests             mpfit.py:1163-1335
//                         (ONE launch per round over five lists: fresh records binned by how many Newton iterations
//                         lmpar needed for that fit last time, and fits parked in the middle of lmpar - see CNT_* below)
// Each kernel takes one queue entry group per block (16 fits in kA, 64 in kB) and appends every fit to the
// queue of the kernel it needs next (accepted step -> kA, rejected step -> kB again, terminated -> done), so
// every wave of every launch is full, whatever the iteration counts are.  Kernel boundaries give the
// inter-workgroup visibility; no data-path collective, no atomics other than one per wave per queue tail.
// Arithmetic: identical, operation for operation, to fsq_lm_core.h (the reference's mpfit order); where an
// operation is evaluated by a cheaper instruction sequence (hoisted-reciprocal division, range-specialised
// 0.5/sqrt) the sequence is the compiler's own with its no-op wrappers removed - see fsq_devmath.h.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include <mutex>
#include <new>
#include <vector>

#include "fsq_common.h"
#include "fsq_lm_quad.h"

namespace {

// LDS hand-offs between the lanes of one wave: keep the compiler from moving LDS accesses across the point
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

#ifdef FSQ_DEBUG_HZ
__device__ unsigned long long g_hz[32];
#endif
#ifdef FSQ_PHASE_PROFILE
__device__ unsigned long long g_rphase[32];
#define RPH_DECL unsigned long long rph_t0 = clock64(), rph_acc[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
#define RPH_MARK(k) { unsigned long long t_ = clock64(); rph_acc[k] += t_ - rph_t0; rph_t0 = t_; }
// (one block in 64 reports: sixteen atomics per wave on two cache lines would themselves be the slowest thing in the kernel -
// 88 atomics per microsecond and line chip-wide, DESIGN.md 4.2 - and the profile would measure its own traffic jam)
#define RPH_FLUSH(off) if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { for (int k_ = 0; k_ < 16; k_++) atomicAdd(&g_rphase[(off) + k_], rph_acc[k_]); }
#else
#define RPH_DECL
#define RPH_MARK(k)
#define RPH_FLUSH(off)
#endif

// ---- data layout in HBM ------------------------------------------------------------------------------
// Work queues are structure-of-arrays indexed by QUEUE POSITION, so a wave reads its 64 (kB) or 16 (kA)
// records with fully coalesced loads; a fit's live state travels with it from queue to queue:
//   queue A record (input of kA):  idx | x[7] | diag[7] | llim1 fnorm par delta xnorm | niter,nfev | E[25]   46 x 8 B
//                                  (E = exp(-(u^2 + v^2) / 2) of the 25 model pixels at x, written by the step round on acceptance:
//                                  kA rebuilds fvec = data - (x0 + x1 E) from it.  Until round 4 E lived in a by-candidate array,
//                                  which the step round - one lane per fit - wrote with 25 stores of 64 scattered 8-byte pieces each)
//   queue B record (input of kB):  the same 21 + gnorm | ipvt | qtf[7] | sdiag[7] | R upper[28]          65 x 8 B
//   queue C record (kB, resumed):  the same 65 + parl paru fp | lmpar iterations done                    69 x 8 B
//   slow queue (plain-division kA): copies of queue-A records of fits that left the guarded operand ranges
// Indexed by candidate: a compact 64-byte copy of the 25 ROI pixels (kinit), the model's 25 exponentials E at the current
// point (written on acceptance, read by kA, which rebuilds fvec = data - (x0 + x1 E) from them), the final result, the ROI
// statistics.
// Counters of one ping/pong set of queues (8 ints): fits waiting in queue A, in the two lists of queue B and the two of queue C.
// Queues B and C each hold TWO lists in one array, one growing from position 0 upwards and one from cap - 1 downwards:
//   B lo / hi   input of the step round, binned by the number of Newton iterations lmpar took for this fit LAST time
//               (<= 1 / more: the count is sticky per fit, oracle statistics in DESIGN.md) - a wave's lanes then mostly finish
//               lmpar together instead of all waiting for the slowest
//   C 1 / 3     fits parked in the middle of lmpar after 1 resp. 3 iterations; they are picked up by the NEXT round's launch
enum { CNT_A = 0, CNT_BLO = 1, CNT_BHI = 2, CNT_C1 = 3, CNT_C3 = 4, CNT_SET = 8 };
enum { Q_EPS = Q_WA3 };           // kA, during qrfac: relative error bounds of the tracked column norms (by logical position)
enum { A_IDX = 0, A_X = 1, A_DIAG = 8, A_LLIM1 = 15, A_FNORM = 16, A_PAR = 17, A_DELTA = 18, A_XNORM = 19, A_ITER = 20,
       A_HDR = 21 /* the part every kind of record starts with */, A_E = 21 /* queue A only: the model's 25 exponentials E at x */, A_LEN = 46,
       B_GNORM = 21, B_IPVT = 22, B_QTF = 23, B_SDIAG = 30, B_R = 37, B_LEN = 65,
       // a fit parked in the middle of lmpar (queue C): the B record as it stands plus the Newton state on par
       C_PARL = 65, C_PARU = 66, C_FP = 67, C_LMIT = 68, C_LEN = 69 };

struct FitOut {               // by candidate: final parameters (written at termination)
    double x[FSQ_NP];
    int status, niter, nfev, pad;
};
struct FitStat {              // by candidate: ROI statistics (kinit)
    double vmax, vmean;
};

// The by-candidate arrays form a POOL of slots; a batch of candidates (one fsq_fit_* call, or one submission to a
// FsqFitQueue) owns a contiguous range of slots while it is in flight, and its queue records carry (slot, ticket).
// The round kernels only ever see slots - fits of several batches share the queues and the launches.
struct Ctx {
    uint16_t* roi;            // [pool][32]: the 25 pixels of every ROI, gathered once by kinit (64 bytes per fit; with pix32 the
                              // same slots hold uint32 words: 128 bytes per fit)
    int pix32;                // FSQ_PIXELS_U32 images: the fit kernels' P32 instantiations
    FitOut* out;              // [pool]
    FitStat* stat;            // [pool]
    long long cap;            // queue capacity (positions)
    long long pool;           // by-candidate slots
    int* err;                 // first invariant a kernel found broken (0: none) - see fsq_guard
    int* slow_total;          // statistics: fits that went through the plain-division kernel
    int* done;                // [FSQ_MAX_TICKETS]: terminated fits per batch in flight
    int tshift;               // a record's tag = slot | ticket << tshift (one 32-bit word: the kernels are at the register limit)
    int wave_prio;            // late rounds: raise the waves' issue priority (they share CUs with another lane's big kernels)
    int force_redo;           // debug: take qrfac's norm re-computation branch at every step (FSQ_DEBUG_FORCE_NORM_RECOMPUTE)
    int force_slow_mod;       // debug: route every fit with idx % mod == 0 through the plain-division kernel
};

// Every queue position / pool slot a kernel is about to WRITE to is checked against the capacity first: the host sizes the
// queues from upper bounds it keeps itself (FsqFitQueue::alive), and if one of those bounds were ever wrong the write would
// land outside the workspace - a memory fault at best, silent damage at worst.  A position out of range is recorded (the
// largest code wins; FsqFitQueue::look turns it into FSQ_EINTERNAL) and replaced by position 0, which is inside.
enum { G_KINIT_POS = 1, G_KINIT_SLOT = 2, G_KA_BLO = 3, G_KA_BHI = 4, G_KA_OVERLAP = 5, G_KA_SLOW = 6, G_KB_C1 = 7, G_KB_C3 = 8, G_KB_A = 9,
       G_KB_BLO = 10, G_KB_BHI = 11 };
FSQ_DEV long long fsq_guard(const Ctx& c, long long v, long long limit, int code)
{
    if ((unsigned long long)v < (unsigned long long)limit) return v;
    atomicMax(c.err, code);
    return 0;
}

// Reserve one queue slot for every lane with `want` set: one atomic per wave (a single counter saturates at
// ~90 atomics/us chip-wide, far below the millions of appends per round), slots in lane order.
FSQ_DEV int wave_reserve(int* counter, bool want)
{
    const unsigned long long m = __ballot(want);
    if (m == 0) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader);
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

// ... and the position checked against the list's capacity - for the lanes that will WRITE there only: a lane that does not
// append gets the counter value behind the wave's reservation, which equals the capacity when the list becomes exactly full
// and is no violation.
FSQ_DEV long long wave_reserve_checked(const Ctx& c, int* counter, bool want, long long limit, int code)
{
    const int at = wave_reserve(counter, want);
    return want ? fsq_guard(c, at, limit, code) : 0;
}

// One batch of candidates: where its pixels come from, which pool slots it owns, where its rows go.
struct BatchArgs {
    const uint16_t* src; const int32_t* cand; int H, W; long long n; int from_image;
    int pix_fmt;              // FSQ_PIXELS_U16 / FSQ_PIXELS_F16 (images only; stand-alone ROIs are uint16)
    long long base;           // first pool slot
    int ticket;
    int no_queue;             // FSQ_MODE_TEXTBOOK_F32: kinit leaves the clipped start in out[slot].x instead of a queue-A record
};

// Count the lanes with `term` set into their batches' done counters: one atomic per wave per batch present.
FSQ_DEV void wave_mark_done(int* done, bool term, int ticket)
{
    unsigned long long m = __ballot(term);
    const int lane = threadIdx.x & 63;
    while (m) {
        const int leader = __ffsll((long long)m) - 1;
        const int t = __shfl(ticket, leader);
        const unsigned long long mm = __ballot(term && ticket == t);
        if (lane == leader) atomicAdd(done + t, __popcll(mm));
        m &= ~mm;
    }
}

// Queue records, the compact ROI copies and the by-candidate E arrays are each read once per round and written once.
// FSQ_NT=1 accesses them with NON-TEMPORAL loads / stores so that this stream (tens of KB per wave) cannot evict the few KB
// of look-up tables (exp, pow's log, sin/cos) from the 32 KB vector L1.  Measured: the tables hit L1 97.6 % of the time
// anyway and the non-temporal accesses are slower (+3 % step time), so the default is plain accesses.
#ifndef FSQ_NT
#define FSQ_NT 0
#endif
typedef unsigned fsq_u4v __attribute__((ext_vector_type(4)));
FSQ_DEV double nt_ld(const double* p) { return FSQ_NT ? __builtin_nontemporal_load(p) : *p; }
FSQ_DEV void nt_st(double* p, double v) { if (FSQ_NT) __builtin_nontemporal_store(v, p); else *p = v; }
FSQ_DEV uint4 nt_ld4(const void* p)
{
    const fsq_u4v v = FSQ_NT ? __builtin_nontemporal_load((const fsq_u4v*)p) : *(const fsq_u4v*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
FSQ_DEV void nt_st4(void* p, uint4 w)
{
    fsq_u4v v; v.x = w.x; v.y = w.y; v.z = w.z; v.w = w.w;
    if (FSQ_NT) __builtin_nontemporal_store(v, (fsq_u4v*)p); else *(fsq_u4v*)p = v;
}
struct NtRef {
    double* p;
    FSQ_DEV operator double() const { return nt_ld(p); }
    FSQ_DEV void operator=(double v) const { nt_st(p, v); }
};
struct NtQ {        // a queue record (structure of arrays: field f of the record lives at p[f * cap])
    double* p;
    FSQ_DEV NtRef operator[](long long i) const { return NtRef{p + i}; }
};
FSQ_DEV NtQ ntq(const double* p) { return NtQ{const_cast<double*>(p)}; }

FSQ_DEV int tag_slot(const Ctx& c, int tag) { return (int)((unsigned)tag & ((1u << c.tshift) - 1u)); }
FSQ_DEV int tag_ticket(const Ctx& c, int tag) { return (int)((unsigned)tag >> c.tshift); }

FSQ_DEV int rpk(int i, int k) { return i * 7 - (i * (i - 1)) / 2 + (k - i); }     // index of (i,k), i <= k, in R upper[28]
FSQ_DEV double pack2(int a, int b) { return __longlong_as_double(((long long)(unsigned)a) | ((long long)b << 32)); }
FSQ_DEV void unpack2(double v, int* a, int* b) { long long u = __double_as_longlong(v); *a = (int)(unsigned)(u & 0xffffffffll); *b = (int)(u >> 32); }

FSQ_DEV void roi_pixels(const BatchArgs& c, long long idx, double* d)
{
    if (c.from_image) {
        const int f = c.cand[3 * idx], h = c.cand[3 * idx + 1], w = c.cand[3 * idx + 2];
        const size_t base = ((size_t)f * c.H + (h - 2)) * c.W + (w - 2);      // (in pixels: fsq_pixel knows the word size)
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int b = 0; b < 5; b++) d[a * 5 + b] = (double)fsq_pixel(c.src, base + (size_t)a * c.W + b, c.pix_fmt);
    } else {
#pragma unroll
        for (int k = 0; k < FSQ_NPIX; k++) d[k] = (double)c.src[idx * FSQ_NPIX + k];
    }
}

// the 25 pixels of fit idx from the compact copy (one 64-byte line instead of five image rows)
FSQ_DEV void roi_compact(const Ctx& c, long long idx, double* d)
{
    if (c.pix32) {
        const uint32_t* s32 = (const uint32_t*)c.roi + (size_t)idx * 32;
#pragma unroll
        for (int k = 0; k < FSQ_NPIX; k++) d[k] = (double)s32[k];
        return;
    }
    const uint16_t* src = c.roi + (size_t)idx * 32;
    const uint4 a = nt_ld4(src), b = nt_ld4(src + 8), e = nt_ld4(src + 16), f = nt_ld4(src + 24);
    const unsigned w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, e.x, e.y, e.z, e.w, f.x, f.y, f.z, f.w};
#pragma unroll
    for (int k = 0; k < FSQ_NPIX; k++) d[k] = (double)((w[k >> 1] >> (16 * (k & 1))) & 0xffffu);
}

__global__ void __launch_bounds__(256) kinit(Ctx c, BatchArgs b, double* __restrict__ QA, int* __restrict__ cntA)
{
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i0 < b.n;
    const long long i = ok ? i0 : 0;            // idle lanes of the last block recompute fit 0 and store nothing
    const long long pos = wave_reserve_checked(c, cntA, ok && !b.no_queue, c.cap, G_KINIT_POS);     // appended behind whatever the queue already holds
    if (b.n <= 0) return;
    const long long slot = fsq_guard(c, b.base + i, c.pool, G_KINIT_SLOT);
    double v[FSQ_NPIX];
    roi_pixels(b, i, v);
    if (ok && c.pix32) {
        uint32_t* dst = (uint32_t*)c.roi + (size_t)slot * 32;
#pragma unroll
        for (int k = 0; k < 32; k++) dst[k] = (k < FSQ_NPIX) ? (unsigned)v[k] : 0u;
    } else if (ok) {
        unsigned w[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            unsigned lo = (2 * k < FSQ_NPIX) ? (unsigned)v[2 * k] : 0u, hi = (2 * k + 1 < FSQ_NPIX) ? (unsigned)v[2 * k + 1] : 0u;
            w[k] = lo | (hi << 16);
        }
        uint16_t* dst = c.roi + (size_t)slot * 32;
        nt_st4(dst, make_uint4(w[0], w[1], w[2], w[3])); nt_st4(dst + 8, make_uint4(w[4], w[5], w[6], w[7]));
        nt_st4(dst + 16, make_uint4(w[8], w[9], w[10], w[11])); nt_st4(dst + 24, make_uint4(w[12], w[13], w[14], w[15]));
    }
    double mx = v[0], isum = 0.0;
#pragma unroll
    for (int k = 0; k < FSQ_NPIX; k++) { mx = v[k] > mx ? v[k] : mx; isum += v[k]; }
#pragma unroll
    for (int pass = 0; pass < FSQ_NPIX; pass++)          // odd-even transposition sort: median = v[12]
#pragma unroll
        for (int k = (pass & 1); k + 1 < FSQ_NPIX; k += 2) {
            double a = v[k], b_ = v[k + 1];
            v[k] = a < b_ ? a : b_;
            v[k + 1] = a < b_ ? b_ : a;
        }
    if (!ok) return;
    const double vmedian = v[12], vmean = isum / 25.0;
    const double llim1 = (mx - vmean) / 3.0;                               // pflib.py:207-209
    double x0[FSQ_NP] = {vmedian, mx, 2.5, 2.5, 1., 1., 0.};               // pflib.py:201-202
    c.stat[slot].vmax = mx; c.stat[slot].vmean = vmean;
#pragma unroll
    for (int k = 0; k < FSQ_NP; k++) {                                     // gaussfitter.py:202-204
        double t = x0[k];
        if (t > fsq_ulim(k) && fsq_qulim(k)) t = fsq_ulim(k);
        if (t < fsq_llim(k, llim1)) t = fsq_llim(k, llim1);
        x0[k] = t;
    }
    if (b.no_queue) {
#pragma unroll
        for (int k = 0; k < FSQ_NP; k++) c.out[slot].x[k] = x0[k];
        return;
    }
    const NtQ q = ntq(QA + pos);
    const long long cap = c.cap;
#pragma unroll
    for (int k = 0; k < FSQ_NP; k++) {
        q[(A_X + k) * cap] = x0[k];
        q[(A_DIAG + k) * cap] = 0.;
    }
    q[A_IDX * cap] = pack2((int)((unsigned)slot | ((unsigned)b.ticket << c.tshift)), 0);       // (second word: lmpar history, none yet)
    q[A_LLIM1 * cap] = llim1; q[A_FNORM * cap] = 0.; q[A_PAR * cap] = 0.; q[A_DELTA * cap] = 0.; q[A_XNORM * cap] = 0.;
    q[A_ITER * cap] = pack2(1, 0);                                         // niter = 1, nfev = 0  (nfev == 0 <=> fresh)
}

#include "fsq_fit_f32.h"

// qrfac's norm down-dating squares a NumPy scalar, i.e. calls libm's pow(t, 2.0) (mpfit.py:1816) - a log, an exp, two
// dependent table look-ups: a fifth of the Jacobian round's time for a number that is only ever COMPARED (pivot choice,
// the re-computation test).  pow(t, 2.0) is the correctly rounded square unless t^2 lies within pow's own error of a
// rounding boundary: e_pow.c bounds its error before the final rounding by ulperr_exp - 0.5 = 0.009 ulp plus
// |y log x| * relerr_log = |2 ln t| * 1.3 * 2^-68 (< 0.002 ulp for t^2 >= 2^-300).  So with sq = RN(t * t) and
// lo = t * t - sq (exact, one fma): |lo| <= 0.485 ulp(sq) and sq not a power of two (where the spacing changes) imply
// pow(t, 2.0) == sq.  fsq_selftest_square checks that implication against fsq_pow2 on the GPU.
FSQ_DEV bool fsq_square_is_pow2(double t, double sq, double lo)
{
    const int ex = fsq_expo(sq);                                    // sq = m * 2^ex, 0.5 <= m < 1: ulp(sq) = 2^(ex - 53)
    const double thr = __builtin_ldexp(0.485, ex - 53);
    const bool ok = (__builtin_fabs(lo) <= thr) && ((fsq_bits(sq) & 0xfffffffffffffull) != 0ull) && (sq >= 0x1p-300) && (sq < 0x1p300);
    return ok || (t == 0);
}

// exp(-(u^2 + v^2) / 2) of one model pixel from the numerators of its two rotated offsets (gaussfitter.py:128-135)
template <bool FAST>
FSQ_DEV double ka_gauss(double nu, double nv, const FsqDivisor& k4, const FsqDivisor& k5, int* em, bool* bad)
{
    if (FAST) { *em = min(*em, fsq_expo(nu)); *em = min(*em, fsq_expo(nv)); }
    const double u = fsq_div_sel<FAST>(nu, k4), v = fsq_div_sel<FAST>(nv, k5);
    const double e = -(u * u + v * v) / 2.;
    return FAST ? fsq_exp_bf(e, bad) : fsq_exp(e);
}

// ---------------------------------------------------------------------------------------------------
// kA: Jacobian round.  block = 64 threads; L lanes per fit (a "group"), 64 / L fits per wave; one trip per block.
//   L = 4: lane c of a group owns columns c and c + 4 (two 25-row columns in registers, 256 VGPRs, 2 waves per SIMD)
//   L = 8: lane c owns column c alone (128 VGPRs and half the LDS per wave: 4 waves per SIMD)
// Slot s of the 8 columns (7 Jacobian columns + the residual vector f, slot 7) lives in lane s % L, register set s / L.
// A lone wave issues one fp64 instruction every 5 (independent) to 9 (dependent) cycles whatever else is free
// (tools/ubench/fma_latency.hip), and this round is one long dependent chain: its throughput is waves in flight per
// wave latency, not instruction count - hence the 8-lane form, which trades ~1.4x the instructions per fit for twice the
// waves.
// FAST = true: divisions by a shared divisor go through fsq_div_by (fsq_devmath.h) and every operand range that
// makes it bit-identical to `/` is checked on the way; a group that leaves those ranges writes nothing and appends
// a copy of its queue-A record to the slow queue SQ, which the FAST = false build (plain divisions, libm pow, same code)
// works off when the host next looks (fits are independent, so a fit may fall a few rounds behind).
// The FAST build also zeroes the queue counters of the next round (nobody reads or appends to them during kA).
// The FAST = false build strides over the slow queue with a small grid.  Memory round trips are kept off the critical
// path: the whole record is fetched up front (lane c takes fields c, c + L, ...; the scalars wait in LDS), the queue-B
// slot is reserved at the top (a group that turns out not to need it leaves a DEAD record, tag -1, which the step round
// skips), and the parameter-only part of fdjac2 runs while the pixels and E (fetched by the slot the record names) are
// on their way.
#pragma push_macro("QL")
#undef QL
#define QL(off, e) lds[((off) + (e)) * G + grp]
template <int G>
FSQ_DEV double kag_dot7(const double* lds, int grp, int off)
{
    double d = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) { double v = QL(off, i); d = fsq_fma(v, v, d); }
    return d;
}
template <int G>
FSQ_DEV double kag_dot25(const double* lds, int grp, int off)
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

// KA_ROWS: a scheduling fence every few rows of the unrolled 25-row loops (with 128 registers per lane the scheduler must not
// pull all 25 LDS reads of a loop to its top); KA_PIN: keeps a running exponent minimum a chain (left to itself the
// optimiser turns it into one tree at the end of the kernel and keeps every numerator alive for it)
#define KA_ROWS(i) do { if (L == 8 && ((i) % 5) == 4) __builtin_amdgcn_sched_barrier(0); } while (0)
#define KA_PIN(v) do { if (FAST) asm volatile("" : "+v"(v)); } while (0)
// a guarded operand range was left / a decision could not be settled: the fit goes to the plain-division kernel.  -DFSQ_DEBUG_HZ
// counts the reasons (fsq_debug_hz; tools/hz_reasons.py)
#ifdef FSQ_DEBUG_HZ
#define KA_HZ(code, cond) do { if (cond) { hz = true; if (FAST) atomicAdd(&g_hz[code], 1ull); } } while (0)
#else
#define KA_HZ(code, cond) do { if (cond) hz = true; } while (0)
#endif
template <bool FAST, int L, bool P32 = false>
__global__ void __launch_bounds__(64, (L == 8 ? 4 : 2)) kA_jacobian(Ctx c, const double* __restrict__ QA, const int* __restrict__ cntA_p,
                                                                    double* __restrict__ QB, int* __restrict__ cnt_cur,
                                                                    double* __restrict__ SQ, int* __restrict__ slow_cnt,
                                                                    int* __restrict__ next_counters)
{
    constexpr int G = 64 / L;                   // fits per wave
    constexpr int NC = 8 / L;                   // columns per lane
    constexpr int MPX = (FSQ_NPIX + L - 1) / L; // pixels per lane when a 25-pixel job is split over the group
    constexpr int MREC = (A_HDR + L - 1) / L;   // record fields per lane (the header; E is fetched by pixel)
    __shared__ double lds[Q_KA_END * G];
    const int lane = threadIdx.x, grp = lane / L, cl = lane % L, gbase = lane - cl;
    const int n7 = FSQ_NP;
    if (c.wave_prio) __builtin_amdgcn_s_setprio(3);
    const int cntA = FAST ? *cntA_p : *slow_cnt;
    if (!FAST && blockIdx.x == 0 && threadIdx.x == 0 && cntA > 0) atomicAdd(c.slow_total, cntA);
    if (FAST && blockIdx.x == 0 && threadIdx.x < CNT_SET) next_counters[threadIdx.x] = 0;
    double col[NC][FSQ_NPIX];
    RPH_DECL
    const long long cap = c.cap;
    const int stride = gridDim.x * G;
    int base = blockIdx.x * G;
    if (base >= cntA) return;
    do {
        RPH_MARK(0)
        const bool active = (base + grp) < cntA;
        const NtQ qa = ntq(QA + (active ? base + grp : 0));
        double rec[MREC];                   // fields cl, cl + L, ... of the 21-field queue-A record
#pragma unroll
        for (int m = 0; m < MREC; m++) { const int f = cl + L * m; rec[m] = (active && f < A_HDR) ? (double)qa[f * cap] : 0.0; }
        bool hz = false, qhz = false;     // FAST: some operand left the range in which fsq_div_by == `/`
        int emin = 0;                     // FAST: smallest exponent among the tracked numerators
        int tag, niter, nfev, hist;
        unpack2(__shfl(rec[A_IDX / L], gbase + A_IDX % L), &tag, &hist);
        unpack2(__shfl(rec[A_ITER / L], gbase + A_ITER % L), &niter, &nfev);
        // this fit's queue-B slot (see above), in the list its lmpar history selects
        const bool b_hi = hist > 1;
        const int at_lo = wave_reserve(cnt_cur + CNT_BLO, active && cl == 0 && !b_hi);
        const int at_hi = wave_reserve(cnt_cur + CNT_BHI, active && cl == 0 && b_hi);
#ifdef FSQ_EXPERIMENT_EXTRA_ATOMICS_KA   // sensitivity of the round to the queue-tail atomics: N more per wave on the same cache line
        {
            int sink = 0;
            for (int x = 0; x < FSQ_EXPERIMENT_EXTRA_ATOMICS_KA; x++) sink += wave_reserve(cnt_cur + 5 + (x & 1), active && cl == 0);
            if (sink == 0x7fffffff) atomicMax(c.err, 99);
        }
#endif
        if (active && cl == 0) {      // (two lists in one array, growing towards each other)
            if (!b_hi) fsq_guard(c, at_lo, cap, G_KA_BLO); else fsq_guard(c, at_hi, cap, G_KA_BHI);
            if ((long long)at_lo + (long long)at_hi + 2 > cap && (at_lo > 0 || at_hi > 0)) {
                const int nlo = b_hi ? cnt_cur[CNT_BLO] : at_lo + 1, nhi = b_hi ? at_hi + 1 : cnt_cur[CNT_BHI];
                if ((long long)nlo + nhi > cap) atomicMax(c.err, (int)G_KA_OVERLAP);
            }
        }
        const long long at = __shfl((active && cl == 0) ? (int)fsq_guard(c, b_hi ? (long long)(cap - 1) - at_hi : (long long)at_lo, cap, b_hi ? G_KA_BHI : G_KA_BLO) : 0, gbase);
        const int idx = tag_slot(c, tag);
        const bool fresh = active && (nfev == 0);
        double llim1 = 0., fnorm = 0., xnorm = 0., delta = 0., par_in = 0.;
        unsigned ipvt = 0x76543210u;
        int status = 0;
        double gnorm = 0.;
        uint4 roi = make_uint4(0u, 0u, 0u, 0u);     // this lane's 32 / L pixels of the compact ROI copy
        uint4 roi_hi = make_uint4(0u, 0u, 0u, 0u);  // (P32: pixels 4..7 of the lane's eight, one 32-bit word each)
        double ev[MPX];
#pragma unroll
        for (int m = 0; m < MPX; m++) ev[m] = 0.0;
        if (active) {                       // pixels and E by pool slot: on their way while the parameters are worked on
            if (P32) {
                static_assert(!P32 || L == 4, "32-bit pixels: the four-lane Jacobian round only");
                const uint32_t* r32 = (const uint32_t*)c.roi + (size_t)idx * 32 + cl * 8;
                roi = nt_ld4(r32); roi_hi = nt_ld4(r32 + 4);
            } else if (L == 4) roi = nt_ld4(c.roi + (size_t)idx * 32 + cl * 8);
            else {
                const unsigned long long w8 = fsq_bits(nt_ld((const double*)(c.roi + (size_t)idx * 32 + cl * 4)));
                roi.x = (unsigned)w8; roi.y = (unsigned)(w8 >> 32);
            }
            if (!fresh) {
#pragma unroll
                for (int m = 0; m < MPX; m++) { const int k = cl + L * m; if (k < FSQ_NPIX) ev[m] = qa[(A_E + k) * cap]; }
            }
#pragma unroll
            for (int m = 0; m < MREC; m++) {    // fields 1..19 -> LDS slots 0..18: x | diag | llim1 fnorm par delta xnorm (Q_X, Q_DIAG, Q_TMP[0..4])
                const int f = cl + L * m;
                if (f >= 1 && f <= 19) QL(Q_X, f - 1) = rec[m];
            }
        }
        WAVE_SYNC();
        RPH_MARK(1)
        if (active) {
            // ---- fdjac2 (mpfit.py:1512-1612): slot s = column s of the Jacobian, slot 7 = f(x) itself -------------
            // The model is g = p0 + p1 * E with E = exp(-(u^2 + v^2) / 2) a function of p2..p6 only, so the columns of p0
            // and p1 and f(x) itself need no new exp: they follow from E at the current point, which is what the step
            // round leaves in the queue-A record (fvec = data - (x0 + x1 * E) is rebuilt from it with the operations that produced it).
            // The five columns that do need the model (c2, c3, sigma4, sigma5, theta) are evaluated split by PIXEL over the
            // group - lane cl takes pixels cl, cl + L, ... for every column - and handed to the lanes that own the columns
            // through the staging slots.
            double xx[FSQ_NP], hh[FSQ_NP];
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) {
                xx[k] = QL(Q_X, k);
                const double eps = 1.4901161193847656e-08;
                double h_ = eps * __builtin_fabs(xx[k]);
                if (h_ == 0) h_ = eps;
                const double ul = k < 2 ? 0.0 : k < 4 ? 3.0 : k < 6 ? 2.0 : 360.0;
                if (k >= 2 && (xx[k] > ul - h_)) h_ = -h_;                   // mpfit.py:1582-1587
                hh[k] = h_;
            }
            bool bad = false;
            double sn, cs;
            fsq_sincos(FSQ_PI_180 * xx[6], &sn, &cs);
            const double rcx = xx[3] * cs - xx[2] * sn, rcy = xx[3] * sn + xx[2] * cs;
            const FsqDivisor k4 = fsq_divisor(xx[4]), k5 = fsq_divisor(xx[5]);
            if (FAST)       // |numerator| <= |p2| + |p3| + 8: bounded once the centre is
                hz = hz || !fsq_divisor_in_range(xx[4]) || !fsq_divisor_in_range(xx[5]) || !(__builtin_fabs(xx[2]) <= 0x1p100) ||
                     !(__builtin_fabs(xx[3]) <= 0x1p100);
            {   // the pixels and E have arrived by now: lane cl converts its 32 / L pixels of the ROI copy
                const unsigned w[4] = {roi.x, roi.y, roi.z, roi.w};
                const unsigned w32[8] = {roi.x, roi.y, roi.z, roi.w, roi_hi.x, roi_hi.y, roi_hi.z, roi_hi.w};
#pragma unroll
                for (int t = 0; t < 32 / L; t++) {
                    const int k = cl * (32 / L) + t;
                    if (k < FSQ_NPIX) QL(Q_DATA, k) = P32 ? (double)w32[t & 7] : (double)((w[t >> 1] >> (16 * (t & 1))) & 0xffffu);
                }
                if (!fresh) {
#pragma unroll
                    for (int m = 0; m < MPX; m++) { const int k = cl + L * m; if (k < FSQ_NPIX) QL(Q_FVEC, k) = ev[m]; }
                }
            }
            WAVE_SYNC();
            if (__ballot(fresh)) {          // mpfit's first function call (mpfit.py:999): E at x0 (whole batches are fresh together)
                if (fresh) {
#pragma unroll 1
                    for (int m = 0; m < MPX; m++) {
                        const int i = cl + L * m;
                        if (i < FSQ_NPIX) {
                            const int xi = i / 5;
                            const double x = (double)xi, y = (double)(i - 5 * xi);
                            const double xp = x * cs - y * sn, yp = x * sn + y * cs;
                            const double E = ka_gauss<FAST>(rcx - xp, rcy - yp, k4, k5, &emin, &bad);
                            QL(Q_FVEC, i) = E;
                        }
                    }
                }
                WAVE_SYNC();
            }
            if (fresh) nfev = 1;
            {   // phase A: the columns of p0, p1 (from E, by their owners) and of the centre (c2, c3: by pixel)
                const double x2p = xx[2] + hh[2], x3p = xx[3] + hh[3];
                const double rcx2 = xx[3] * cs - x2p * sn, rcy2 = xx[3] * sn + x2p * cs;
                const double rcx3 = x3p * cs - xx[2] * sn, rcy3 = x3p * sn + xx[2] * cs;
                if (FAST) KA_HZ(0, !(__builtin_fabs(x2p) <= 0x1p100) || !(__builtin_fabs(x3p) <= 0x1p100));
#pragma unroll 1
                for (int m = 0; m < MPX; m++) {
                    const int i = cl + L * m;
                    if (i < FSQ_NPIX) {
                        const int xi = i / 5;
                        const double x = (double)xi, y = (double)(i - 5 * xi);
                        const double xp = x * cs - y * sn, yp = x * sn + y * cs;
                        const double d = QL(Q_DATA, i);
                        const double E2 = ka_gauss<FAST>(rcx2 - xp, rcy2 - yp, k4, k5, &emin, &bad);
                        const double E3 = ka_gauss<FAST>(rcx3 - xp, rcy3 - yp, k4, k5, &emin, &bad);
                        QL(Q_STAGE, i) = d - (xx[0] + xx[1] * E2);
                        QL(Q_STAGE + 25, i) = d - (xx[0] + xx[1] * E3);
                    }
                }
                const double p0 = (cl == 0) ? xx[0] + hh[0] : xx[0], p1 = (cl == 1) ? xx[1] + hh[1] : xx[1];
#pragma unroll
                for (int i = 0; i < FSQ_NPIX; i++) { col[0][i] = QL(Q_DATA, i) - (p0 + p1 * QL(Q_FVEC, i)); KA_ROWS(i); }    // slots 0, 1 (lanes 0, 1)
                WAVE_SYNC();
                if (cl == 2 || cl == 3) {                                                           // slots 2, 3
#pragma unroll
                    for (int i = 0; i < FSQ_NPIX; i++) { col[0][i] = QL(Q_STAGE + 25 * (cl - 2), i); KA_ROWS(i); }
                }
                // every lane has read E: fvec = f(x) takes its place (each lane its own pixels)
                double fvm[MPX];
#pragma unroll
                for (int m = 0; m < MPX; m++) {
                    const int i = cl + L * m;
                    fvm[m] = (i < FSQ_NPIX) ? QL(Q_DATA, i) - (xx[0] + xx[1] * QL(Q_FVEC, i)) : 0.0;
                }
                WAVE_SYNC();
#pragma unroll
                for (int m = 0; m < MPX; m++) {
                    const int i = cl + L * m;
                    if (i < FSQ_NPIX) QL(Q_FVEC, i) = fvm[m];
                }
            }
            {   // phase B: the columns of sigma4, sigma5 (one new quotient each) and theta (a new rotation), by pixel
                const double x4p = xx[4] + hh[4], x5p = xx[5] + hh[5];
                const FsqDivisor k4p = fsq_divisor(x4p), k5p = fsq_divisor(x5p);
                double snt, cst;
                fsq_sincos(FSQ_PI_180 * (xx[6] + hh[6]), &snt, &cst);
                const double rcxt = xx[3] * cst - xx[2] * snt, rcyt = xx[3] * snt + xx[2] * cst;
                if (FAST) KA_HZ(1, !fsq_divisor_in_range(x4p) || !fsq_divisor_in_range(x5p));
#pragma unroll 1
                for (int m = 0; m < MPX; m++) {
                    const int i = cl + L * m;
                    if (i < FSQ_NPIX) {
                        const int xi = i / 5;
                        const double x = (double)xi, y = (double)(i - 5 * xi);
                        const double xp = x * cs - y * sn, yp = x * sn + y * cs;
                        const double nu = rcx - xp, nv = rcy - yp;
                        if (FAST) { emin = min(emin, fsq_expo(nu)); emin = min(emin, fsq_expo(nv)); }
                        const double ub = fsq_div_sel<FAST>(nu, k4), vb = fsq_div_sel<FAST>(nv, k5);
                        const double u4 = fsq_div_sel<FAST>(nu, k4p), v5 = fsq_div_sel<FAST>(nv, k5p);
                        const double e4 = -(u4 * u4 + vb * vb) / 2., e5 = -(ub * ub + v5 * v5) / 2.;
                        const double E4 = FAST ? fsq_exp_bf(e4, &bad) : fsq_exp(e4);
                        const double E5 = FAST ? fsq_exp_bf(e5, &bad) : fsq_exp(e5);
                        const double xpt = x * cst - y * snt, ypt = x * snt + y * cst;
                        const double Et = ka_gauss<FAST>(rcxt - xpt, rcyt - ypt, k4, k5, &emin, &bad);
                        const double d = QL(Q_DATA, i);
                        QL(Q_STAGE, i) = d - (xx[0] + xx[1] * E4);
                        QL(Q_STAGE + 25, i) = d - (xx[0] + xx[1] * E5);
                        QL(Q_STAGE + 50, i) = d - (xx[0] + xx[1] * Et);
                    }
                }
                WAVE_SYNC();
                // slots 4, 5, 6 and 7 (= fvec): lanes 0..3 second register set (L = 4), lanes 4..7 (L = 8)
                const int s47 = (L == 4) ? cl + 4 : cl;
                if (s47 >= 4) {
#pragma unroll
                    for (int i = 0; i < FSQ_NPIX; i++) { col[NC - 1][i] = QL((s47 < 7) ? Q_STAGE + 25 * (s47 - 4) : Q_FVEC, i); KA_ROWS(i); }
                }
            }
            if (FAST) KA_HZ(2, bad);
            double hcol[NC];
#pragma unroll
            for (int pass = 0; pass < NC; pass++) {
                const int slot = cl + L * pass;
                double h_ = 0.0;
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) h_ = (slot == k) ? hh[k] : h_;
                hcol[pass] = h_;                  // 0 for slot 7: its quotient is not used
            }
            RPH_MARK(2)
            if (fresh && cl == 0) QL(Q_TMP, 1) = fsq_sqrt(kag_dot25<G>(lds, grp, Q_FVEC));     // fnorm of a fresh fit
            nfev += 7;
            WAVE_SYNC();            // (the staging slots are free again: qrfac writes Q_ACN .. Q_R below)
            {
                double ssum[NC];
                FsqDivisor kh[NC];
#pragma unroll
                for (int pass = 0; pass < NC; pass++) {
                    ssum[pass] = 0.0;
                    kh[pass] = fsq_divisor(hcol[pass]);
                    if (FAST) KA_HZ(3, !fsq_divisor_in_range(hcol[pass]));
                }
                // residuals are bounded by 2^16 + x0 + x1, so the numerators below stay under 2^102
                if (FAST) KA_HZ(4, !(QL(Q_X, 0) <= 0x1p100) || !(QL(Q_X, 1) <= 0x1p100));
#pragma unroll
                for (int i = 0; i < FSQ_NPIX; i++) {
                    const double fv = QL(Q_FVEC, i);
#pragma unroll
                    for (int pass = 0; pass < NC; pass++) {
                        const int slot = cl + L * pass;
                        const double nn = col[pass][i] - fv;
                        if (FAST) { emin = min(emin, fsq_expo(nn)); KA_PIN(emin); }
                        const double qq = fsq_div_sel<FAST>(nn, kh[pass]);
                        col[pass][i] = (slot < 7) ? qq : fv;
                        ssum[pass] += fv * col[pass][i];
                    }
                    KA_ROWS(i);
                }
#pragma unroll
                for (int pass = 0; pass < NC; pass++) {     // pegged parameters (mpfit.py:1073-1091)
                    const int slot = cl + L * pass;
                    bool peg = false;
                    if (slot < 7) {
                        const double xs = QL(Q_X, slot), ll1 = QL(Q_TMP, 0);
                        const bool lp = (xs == fsq_llim(slot, ll1)), up = fsq_qulim(slot) && (xs == fsq_ulim(slot));
                        peg = (lp && ssum[pass] > 0) || (up && ssum[pass] < 0);
                    }
#pragma unroll
                    for (int i = 0; i < FSQ_NPIX; i++)
                        if (peg) col[pass][i] = 0;
                }
            }
            RPH_MARK(3)
            // ---- qrfac with column pivoting (mpfit.py:1748-1822), Q^T f fused in as slot 7 ----------
            // Round 4 (production instantiation: FAST, 4 lanes): the tracked column norms, their reference values and their error
            // bounds - rdiag, wa, eps, indexed by LOGICAL POSITION - live in registers, lane cl of the quad holding positions cl
            // and cl + 4, instead of in LDS where every lane read all of them for every pivot choice (the pivot search was 14 % of
            // the kernel's time for 4 % of its instructions: round trips to LDS in a dependent chain).  Maxima and the first
            // position that attains them are quad reductions; a position's value is fetched from its owner with one shuffle.
            constexpr bool REGPOS = FAST && (L == 4);
            double prd[2] = {0., 0.}, pwa[2] = {0., 0.}, pep[2] = {0., 0.};
            auto pos_get = [&](double a0, double a1, int p) { return __shfl((p >> 2) ? a1 : a0, gbase + (p & 3)); };
#pragma unroll
            for (int pass = 0; pass < NC; pass++) {
                const int slot = cl + L * pass;
                if (slot < 7) {
                    const double nn = fsq_sqrt(dot_regcol(col[pass], 25));
                    QL(Q_ACN, slot) = nn;
                    if (REGPOS) { prd[pass] = nn; pwa[pass] = nn; pep[pass] = 0.; }
                    else { QL(Q_RDIAG, slot) = nn; QL(Q_WA, slot) = nn; QL(Q_EPS, slot) = 0.; }
                }
            }
            WAVE_SYNC();
            unsigned pos = 0x76543210u;         // slot -> position
            bool broken = false;
            for (int j = 0; j < n7; j++) {
                const int len = FSQ_NPIX - j;
                if (REGPOS && !broken) {
                    // candidates: positions j .. 6 (norms are >= +0; a NaN among them sends the fit to the exact kernel, where
                    // numpy.max's order of comparisons is followed literally)
                    double lm = -1.0;
                    bool has_nan = false;
#pragma unroll
                    for (int m = 0; m < 2; m++) {
                        const int p = cl + 4 * m;
                        if (p >= j && p < n7) { has_nan = has_nan || (prd[m] != prd[m]); lm = (prd[m] > lm) ? prd[m] : lm; }
                    }
                    double rmax = lm;
                    { const double o1 = __shfl_xor(rmax, 1); rmax = (o1 > rmax) ? o1 : rmax; const double o2 = __shfl_xor(rmax, 2); rmax = (o2 > rmax) ? o2 : rmax; }
                    int lk = 99;
#pragma unroll
                    for (int m = 1; m >= 0; m--) {
                        const int p = cl + 4 * m;
                        if (p >= j && p < n7 && prd[m] == rmax) lk = p;
                    }
                    int kmax = min(lk, __shfl_xor(lk, 1));
                    kmax = min(kmax, __shfl_xor(kmax, 2));
                    if (kmax == 99) kmax = -1;
                    KA_HZ(5, kmax < 0 || has_nan);
                    const int kq = kmax < 0 ? j : kmax;
                    const double em = pos_get(pep[0], pep[1], kq), low_m = rmax * (1. - em);
#pragma unroll
                    for (int m = 0; m < 2; m++) {
                        const int p = cl + 4 * m;
                        if (p >= j && p < n7 && p != kq) KA_HZ(6, (pep[m] != 0 || em != 0) && !(prd[m] * (1. + pep[m]) < low_m));
                    }
                    const double vj_rd = pos_get(prd[0], prd[1], j), vj_wa = pos_get(pwa[0], pwa[1], j), vj_ep = pos_get(pep[0], pep[1], j);
                    if (kmax >= 0 && kmax != j) {
                        int sj = nib_get(ipvt, j), sk = nib_get(ipvt, kmax);
                        ipvt = nib_set(nib_set(ipvt, j, sk), kmax, sj);
                        pos = nib_set(nib_set(pos, sk, j), sj, kmax);
#pragma unroll
                        for (int m = 0; m < 2; m++)
                            if (cl + 4 * m == kmax) { prd[m] = vj_rd; pwa[m] = vj_wa; pep[m] = vj_ep; }
                    }
                }
                if (!REGPOS && !broken) {
                    double rmax = QL(Q_RDIAG, j);
                    for (int k = j + 1; k < n7; k++) rmax = np_max2(rmax, QL(Q_RDIAG, k));
                    int kmax = -1;
                    for (int k = n7 - 1; k >= j; k--)
                        if (QL(Q_RDIAG, k) == rmax) kmax = k;
                    if (FAST) {
                        // the tracked norms carry error bounds (see the down-dating below): the choice is the reference's
                        // when every other candidate lies clearly below the chosen one (exact values compare exactly)
                        KA_HZ(5, kmax < 0);
                        if (kmax >= 0) {
                            const double em = QL(Q_EPS, kmax), low_m = rmax * (1. - em);
                            for (int k = j; k < n7; k++) {
                                const double ek = QL(Q_EPS, k);
                                KA_HZ(6, k != kmax && (ek != 0 || em != 0) && !(QL(Q_RDIAG, k) * (1. + ek) < low_m));
                            }
                        }
                    }
                    if (kmax >= 0 && kmax != j) {
                        int sj = nib_get(ipvt, j), sk = nib_get(ipvt, kmax);
                        ipvt = nib_set(nib_set(ipvt, j, sk), kmax, sj);
                        pos = nib_set(nib_set(pos, sk, j), sj, kmax);
                        QL(Q_RDIAG, kmax) = QL(Q_RDIAG, j);
                        QL(Q_WA, kmax) = QL(Q_WA, j);
                        if (FAST) QL(Q_EPS, kmax) = QL(Q_EPS, j);
                    }
                }
                RPH_MARK(7)
                const int lj = nib_get(ipvt, j);
                const int owner = gbase + (lj % L);
                // The lane that owns the pivot column publishes it in LDS (the Q_DATA slots: the pixels are not needed again
                // in this pass) together with its norm; the group then turns it into the Householder vector row-parallel
                // (lane cl scales rows cl, cl + L, ...) and every lane reads the vector from there, so no lane keeps a third
                // 25-row column in registers and the scaling is not one lane's work.  (Letting every lane take the norm from
                // the published column instead - no select between the owner's register sets - was measured 2 % slower.)
                bool brk = false;
                if (lane == owner) {
                    double t[FSQ_NPIX];
#pragma unroll
                    for (int i = 0; i < FSQ_NPIX; i++) t[i] = (NC == 2 && (lj / L) != 0) ? col[NC - 1][i] : col[0][i];
#pragma unroll
                    for (int i = 0; i < FSQ_NPIX; i++) { QL(Q_DATA, i) = t[i]; KA_ROWS(i); }
                    if (!broken) {
                        double ajnorm = fsq_sqrt(dot_regcol(t, len));
                        if (ajnorm == 0) brk = true;                // mpfit.py:1790 `break`
                        else {
                            if (t[0] < 0) ajnorm = -ajnorm;
                            QL(Q_TMP, 5) = -ajnorm;
                            QL(Q_TMP, 6) = ajnorm;
                        }
                    }
                }
                WAVE_SYNC();
                broken = broken || (__shfl((int)brk, owner) != 0);
                int emin_s = 0;                                     // FAST: lower bound of the scaled reflector's exponents
                if (!broken) {
                    const double ajn = QL(Q_TMP, 6);
                    const FsqDivisor kn = fsq_divisor(ajn);
                    int er = 0;
                    if (FAST) KA_HZ(7, !fsq_divisor_in_range(ajn));
#pragma unroll
                    for (int m = 0; m < MPX; m++) {
                        const int i = cl + L * m;
                        if (i < FSQ_NPIX) {
                            const double raw = QL(Q_DATA, i);       // rows >= len are zeros
                            if (FAST) { er = min(er, fsq_expo(raw)); KA_PIN(er); }
                            double q_ = fsq_div_sel<FAST>(raw, kn);
                            if (i == 0) q_ = q_ + 1;
                            QL(Q_DATA, i) = q_;
                        }
                    }
                    if (FAST) {                                     // |raw| <= |ajnorm|: no upper check needed
#pragma unroll
                        for (int d = 1; d < L; d <<= 1) er = min(er, __shfl_xor(er, d));
                        emin = min(emin, er);
                        emin_s = er - fsq_expo(ajn) - 1;
                    }
                }
                WAVE_SYNC();
                RPH_MARK(8)
#define REFL(i) QL(Q_DATA, i)
                const double ajj0 = REFL(0);
                const FsqDivisor kj = fsq_divisor(ajj0);
                if (FAST) KA_HZ(8, !fsq_divisor_in_range(ajj0));
#pragma unroll
                for (int pass = 0; pass < NC; pass++) {
                    const int slot = cl + L * pass;
                    const bool is_f = (slot == 7);
                    const int k = is_f ? 7 : nib_get(pos, slot);
                    const bool todo = is_f ? true : (!broken && k > j);
                    if (todo && ajj0 != 0) {
                        double s = 0.0;
                        // rows >= len hold zeros (see the shift below): they add +0 to the sum and stay zero in
                        // the update, so neither loop needs a bound check and both are straight-line code
#pragma unroll
                        for (int i = 0; i < FSQ_NPIX; i++) { s += col[pass][i] * REFL(i); KA_ROWS(i); }
                        if (FAST) {     // numerators REFL(i) * s: |REFL(i)| < 4, exponent >= emin_s (or zero)
                            const int es = fsq_expo(s);
                            KA_HZ(9, (emin_s + es - 2 < -FSQ_DIV_EN) || (es + 2 > FSQ_DIV_EN));
                        }
#pragma unroll
                        for (int i = 0; i < FSQ_NPIX; i++) { col[pass][i] = col[pass][i] - fsq_div_sel<FAST>(REFL(i) * s, kj); KA_ROWS(i); }
                    }
                    if (is_f) QL(Q_QTF, j) = col[pass][0];
                    else if (nib_get(pos, slot) > j) QL(Q_R, j * 7 + slot) = col[pass][0];
                }
                // Norm down-dating of the live columns (mpfit.py:1810-1820).  It is scalar work per column, and at step j only
                // the 6 - j columns at positions > j are live, scattered over the 8 slots of the group: the lanes take the live
                // POSITIONS in order - lane cl position j + 1 + cl (then j + 1 + L + cl) - reading the column's new leading
                // element R(j, .) from LDS.  The rare full recomputation of a norm needs the column itself, so that goes back
                // to the lane that holds it.
                WAVE_SYNC();
                RPH_MARK(9)
                constexpr int NSUB = (L == 4) ? 2 : 1;
                unsigned long long redo[NSUB];
#pragma unroll
                for (int sub = 0; sub < NSUB; sub++) {
                    // (register-resident norms: lane cl looks after the positions it holds, cl and cl + 4; else the live positions
                    // are dealt out in order, lane cl taking j + 1 + cl and j + 1 + L + cl)
                    const int p = REGPOS ? cl + 4 * sub : j + 1 + cl + L * sub;
                    bool need = false;
                    if (p > j && p < n7 && !broken && ajj0 != 0) {
                        double rk = REGPOS ? prd[sub] : (double)QL(Q_RDIAG, p);
                        if (rk != 0 && !FAST) {
                            double temp = QL(Q_R, j * 7 + nib_get(ipvt, p)) / rk;
                            rk = rk * fsq_sqrt(np_max2(1. - fsq_pow2(temp), 0.));
                            temp = rk / QL(Q_WA, p);
                            if ((0.05 * temp * temp) <= FSQ_MACHEP || c.force_redo) need = true;
                            else QL(Q_RDIAG, p) = rk;
                        }
                        if (rk != 0 && FAST) {
                            // The same down-dating with the square taken by one multiplication.  Where that provably is
                            // pow(t, 2.0) (fsq_square_is_pow2) and the norm going in was exact, every operation below is the
                            // reference's and the result is exact again (bound 0).  Otherwise the result carries a relative
                            // error bound e1 against the reference's value, which the two decisions these norms feed - the
                            // re-computation test here and the pivot choice of the next steps - take into account; a decision
                            // the bound cannot settle sends the fit to the exact kernel (hz), like every other guarded range.
                            const double U = 1.1102230246251565e-16;
                            const double e0 = REGPOS ? pep[sub] : (double)QL(Q_EPS, p);
                            const double t = QL(Q_R, j * 7 + nib_get(ipvt, p)) / rk;
                            const double sq = t * t, lo = fsq_fma(t, t, -sq);
                            const bool exact = (e0 == 0) && fsq_square_is_pow2(t, sq, lo);
                            const double v = 1. - sq;
                            const double dv = exact ? 0. : sq * (2. * e0 + 8. * U) + 2. * U;      // bound of |v - v_ref|
                            KA_HZ(10, !(v == v));
                            KA_HZ(11, !exact && !(__builtin_fabs(v) > dv));                     // sign of 1 - t^2 not settled
                            const double rk1 = rk * fsq_sqrt(np_max2(v, 0.));
                            double e1 = 0.;
                            if (!exact && v > 0) e1 = e0 + 0.6 * dv * __builtin_amdgcn_rcp(v) + 4. * U;
                            KA_HZ(12, !(e1 < 1e-3));
                            const double temp = rk1 / (REGPOS ? pwa[sub] : (double)QL(Q_WA, p));
                            const double q = 0.05 * temp * temp;
                            if (e1 == 0) need = (q <= FSQ_MACHEP);
                            else {
                                const double qe = 2. * e1 + 8. * U;
                                if (q * (1. + qe) <= FSQ_MACHEP) need = true;
                                else KA_HZ(14, !(q * (1. - qe) > FSQ_MACHEP));
                            }
                            if (c.force_redo) need = true;
                            if (!need) {
                                if (REGPOS) { prd[sub] = rk1; pep[sub] = e1; }
                                else { QL(Q_RDIAG, p) = rk1; QL(Q_EPS, p) = e1; }
                            }
                        }
                    }
                    redo[sub] = __ballot(need);
                }
                unsigned long long any_redo = 0ull;
#pragma unroll
                for (int sub = 0; sub < NSUB; sub++) any_redo |= redo[sub];
                if (any_redo) {
#pragma unroll
                    for (int pass = 0; pass < NC; pass++) {
                        const int slot = cl + L * pass;
                        const int k = (slot < 7) ? nib_get(pos, slot) : 0;
                        const int ix = REGPOS ? k : k - j - 1;      // (which lane looked at position k, and in which of its two turns)
                        if (slot < 7 && k > j && ((redo[(NSUB == 2) ? ix / L : 0] >> (gbase + ix % L)) & 1ull)) {
                            const double rk = fsq_sqrt(dot_regcol_from1(col[pass], len));
                            QL(Q_WA, k) = rk;
                            QL(Q_RDIAG, k) = rk;
                            if (FAST) QL(Q_EPS, k) = 0.;            // an exact norm again
                        }
                    }
                    if (REGPOS) {       // (rare: the column's owner has left the fresh norm in LDS for the lane that holds the position)
                        WAVE_SYNC();
#pragma unroll
                        for (int sub = 0; sub < 2; sub++)
                            if ((redo[sub] >> lane) & 1ull) { prd[sub] = QL(Q_WA, cl + 4 * sub); pwa[sub] = prd[sub]; pep[sub] = 0.; }
                    }
                }
                RPH_MARK(10)
                if (REGPOS) {
                    double dj = QL(Q_TMP, 5);
                    if (broken) dj = pos_get(prd[0], prd[1], j);
                    QL(Q_R, j * 7 + lj) = dj;                       // fjac[j, lj] = rdiag[j] (mpfit.py:1123)
                } else {
                    if (!broken) QL(Q_RDIAG, j) = QL(Q_TMP, 5);
                    QL(Q_R, j * 7 + lj) = QL(Q_RDIAG, j);           // fjac[j, lj] = rdiag[j] (mpfit.py:1123)
                }
#pragma unroll
                for (int pass = 0; pass < NC; pass++) {
#pragma unroll
                    for (int i = 0; i + 1 < FSQ_NPIX; i++) col[pass][i] = col[pass][i + 1];
                    col[pass][FSQ_NPIX - 1] = 0.0;
                }
                WAVE_SYNC();
                RPH_MARK(11)
            }
            RPH_MARK(4)
            // ---- first iteration scaling, gradient test (mpfit.py:1099-1160) -----------------------
#define QRG(i, k) QL(Q_R, (i) * 7 + nib_get(ipvt, (k)))
            llim1 = QL(Q_TMP, 0); fnorm = QL(Q_TMP, 1); par_in = QL(Q_TMP, 2); delta = QL(Q_TMP, 3); xnorm = QL(Q_TMP, 4);
            if (niter == 1) {
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) {
                    double a = QL(Q_ACN, k);
                    double dg = (a == 0) ? 1. : a;
                    QL(Q_DIAG, k) = dg;
                    QL(Q_WA3, k) = dg * QL(Q_X, k);
                }
                xnorm = fsq_sqrt(kag_dot7<G>(lds, grp, Q_WA3));
                delta = 100. * xnorm;
                if (delta == 0.) delta = 100.;
            }
            gnorm = 0.;
            if (fnorm != 0) {
                for (int j = 0; j < n7; j++) {
                    double an = QL(Q_ACN, nib_get(ipvt, j));
                    if (an != 0) {
                        double sg = 0.0;
                        for (int i = 0; i <= j; i++) sg += QRG(i, j) * QL(Q_QTF, i);
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
            RPH_MARK(5)
            // ---- finish it (gradient test) ... ---------------------------------------------------------------
            if (FAST) {
                // one verdict per fit: any lane out of range sends the whole fit to the plain-division kernel
                KA_HZ(13, (emin < -FSQ_DIV_EN));
                if (c.force_slow_mod > 0 && (idx % c.force_slow_mod) == 0) hz = true;
                const unsigned long long m = __ballot(hz);
                qhz = ((m >> gbase) & ((1ull << L) - 1ull)) != 0;
            }
            if (status != 0 && cl == 0 && !qhz) {
                FitOut o;
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) o.x[k] = QL(Q_X, k);
                o.status = status; o.niter = niter; o.nfev = nfev; o.pad = 0;
                c.out[idx] = o;
            }
        }
        wave_mark_done(c.done, active && status != 0 && cl == 0 && !qhz, tag_ticket(c, tag));
        // ---- ... or hand it over to the step round (the queue-B slot reserved at the top) --------------------------------
        {
            bool go = active && (status == 0);
            if (FAST) {
                int sat = (int)wave_reserve_checked(c, slow_cnt, qhz && cl == 0, cap, G_KA_SLOW);
                sat = __shfl(sat, gbase);
                if (qhz) for (int f = cl; f < A_LEN; f += L) nt_st(SQ + (size_t)sat + f * cap, qa[f * cap]);
                go = go && !qhz;
            }
            const NtQ qb = ntq(QB + at);
            if (go) {
                for (int e = cl; e < 28; e += L) {
                    int i = 0, rem = e;
                    while (rem >= 7 - i) { rem -= 7 - i; i++; }
                    qb[(B_R + e) * cap] = QRG(i, i + rem);
                }
                for (int k = cl; k < FSQ_NP; k += L) {
                    qb[(A_X + k) * cap] = QL(Q_X, k);
                    qb[(A_DIAG + k) * cap] = QL(Q_DIAG, k);
                    qb[(B_QTF + k) * cap] = QL(Q_QTF, k);
                    qb[(B_SDIAG + k) * cap] = 0.;
                }
                if (cl == 0) {
                    qb[A_IDX * cap] = pack2(tag, hist);
                    qb[A_LLIM1 * cap] = llim1; qb[A_FNORM * cap] = fnorm; qb[A_PAR * cap] = par_in; qb[A_DELTA * cap] = delta;
                    qb[A_XNORM * cap] = xnorm; qb[A_ITER * cap] = pack2(niter, nfev);
                    qb[B_GNORM * cap] = gnorm; qb[B_IPVT * cap] = pack2((int)ipvt, 0);
                }
            } else if (active && cl == 0) qb[A_IDX * cap] = pack2(-1, 0);        // terminated or sent to the slow queue: a dead slot
        }
        RPH_MARK(6)
        base += stride;
    } while (!FAST && base < cntA);
    RPH_FLUSH(16)
}
#undef QRG
#undef REFL
#undef KA_ROWS
#undef KA_PIN
#undef KA_HZ
#pragma pop_macro("QL")

// ---------------------------------------------------------------------------------------------------
// kB: step round.  One lane per fit, one 64-fit tile per block.
// lmpar's Newton iteration on par takes 0 or 1 iterations for half of the fits, 3 for a fifth and all 10 for a quarter of them
// - and whatever a fit needed last time it most likely needs again (0 -> 0: 90 %, 1 -> 1: 86 %, 3 -> 3: 75 %, 10 -> 10: 82 %).
// A wave waits for its slowest lane, so the fits are kept apart by that history: queue B holds two lists (last count <= 1 / more),
// whose tiles stop lmpar after 1 resp. 3 iterations and PARK the lanes that are not done (their R / sdiag / par state as it
// stands) in queue C, again in two lists (parked after 1 / after 3 iterations); the next round's launch picks those up and
// runs them to 3 resp. to the end.  One launch per round works off all four lists.
#ifndef FSQ_KA_LANES_DEFAULT
#define FSQ_KA_LANES_DEFAULT 4
#endif
#ifndef FSQ_LMPAR_FIRST
#define FSQ_LMPAR_FIRST 5
#endif
#ifndef FSQ_LMPAR_LO
#define FSQ_LMPAR_LO 1
#endif
#ifndef FSQ_KB_WAVES
#define FSQ_KB_WAVES 2
#endif
// The trial evaluation f(wa2) of the step round (mpfit.py:1245): E = exp(-(u^2 + v^2) / 2) of the 25 model pixels into the
// lane's LDS column (the residuals data - (p0 + p1 * E) are formed by the caller; E is also what an accepted step leaves
// in the queue-A record for the next Jacobian round).
// FAST: the model's two divisions per pixel share their divisors (sigma_h, sigma_w, inside [0.75, 2] by the bounds) ->
// fsq_div_by, and exp is the branch-free fsq_exp_bf; the operand ranges in which those equal `/` and exp() bit for bit
// are checked on the way and the return value is true when one was left - the caller then repeats the evaluation with
// FAST = false (plain divisions, full exp; same order of operations as fsq_model, gaussfitter.py:100-136).
// Rows are a rolled loop (a fifth of the code) writing to LDS slots - no dynamically indexed register array, no scratch.
FSQ_DEV int kb_res_slot(int i) { return i < 14 ? i : i + 7; }         // slots 14..20 hold the step vector wa1

template <bool FAST>
FSQ_DEV bool kb_trial_gauss(const double* p, double* myscr)
{
    bool bad = false;
    int em = 0;
    double s, c;
    fsq_sincos(FSQ_PI_180 * p[6], &s, &c);
    const double rcen_x = p[3] * c - p[2] * s;
    const double rcen_y = p[3] * s + p[2] * c;
    const FsqDivisor k4 = fsq_divisor(p[4]), k5 = fsq_divisor(p[5]);
    if (FAST)       // |numerator| <= |p2| + |p3| + 8: bounded once the centre is
        bad = !fsq_divisor_in_range(p[4]) || !fsq_divisor_in_range(p[5]) || !(__builtin_fabs(p[2]) <= 0x1p100) ||
              !(__builtin_fabs(p[3]) <= 0x1p100);
#pragma unroll 1
    for (int xi = 0; xi < 5; xi++) {
        const double x = (double)xi;
#pragma unroll
        for (int yi = 0; yi < 5; yi++) {
            const double y = (double)yi;
            const double xp = x * c - y * s;
            const double yp = x * s + y * c;
            const double nu = rcen_x - xp, nv = rcen_y - yp;
            if (FAST) { em = min(em, fsq_expo(nu)); em = min(em, fsq_expo(nv)); }
            const double u = fsq_div_sel<FAST>(nu, k4);
            const double v = fsq_div_sel<FAST>(nv, k5);
            const double e = -(u * u + v * v) / 2.;
            myscr[kb_res_slot(xi * 5 + yi) * 64] = FAST ? fsq_exp_bf(e, &bad) : fsq_exp(e);
        }
    }
    if (FAST) bad = bad || (em < -FSQ_DIV_EN);
    return bad;
}

FSQ_DEV double kb_late_load(const NtRef r) { asm volatile("" ::: "memory"); return (double)r; }

struct KbLimits { int lim[4]; };        // lmpar iteration limits of the four kinds of tiles: B lo, B hi, C 1, C 3
template <bool ALIASED, bool P32 = false>
__global__ void __launch_bounds__(64, FSQ_KB_WAVES) kB_step(Ctx c, const double* __restrict__ QB, const double* __restrict__ QC,
                                                  const int* __restrict__ cnt_cur,
                                                  double* __restrict__ QA_next, double* __restrict__ QB_next, double* __restrict__ QC_next,
                                                  int* __restrict__ cnt_next, KbLimits lims)
{
    __shared__ double scr[32 * 64];
    if (c.wave_prio) __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x;
    const long long cap = c.cap;
    double* myscr = scr + lane;
    RPH_DECL
    {
        // One launch works off all four input lists, one 64-fit tile per block, the long-running kinds first: fits parked after
        // 3 iterations (they run up to 7 more), B hi (3 iterations, then parked), fits parked after 1 iteration, B lo.
        int b = blockIdx.x, seg, cnt;
        const int n_lo = cnt_cur[CNT_BLO], n_hi = cnt_cur[CNT_BHI], n_c1 = cnt_cur[CNT_C1], n_c3 = cnt_cur[CNT_C3];
        if (b < (n_c3 + 63) / 64) { seg = 3; cnt = n_c3; }
        else if ((b -= (n_c3 + 63) / 64) < (n_hi + 63) / 64) { seg = 1; cnt = n_hi; }
        else if ((b -= (n_hi + 63) / 64) < (n_c1 + 63) / 64) { seg = 2; cnt = n_c1; }
        else if ((b -= (n_c1 + 63) / 64) < (n_lo + 63) / 64) { seg = 0; cnt = n_lo; }
        else return;
        const bool resume = seg >= 2;               // (wave-uniform)
        const int lm_limit = lims.lim[seg];
        const int base = b * 64;
        RPH_MARK(0)
        bool live = (base + lane) < cnt;
        const int p_in = live ? (base + lane) : base;
        const NtQ qb = ntq((resume ? QC : QB) + ((seg & 1) ? cap - 1 - p_in : p_in));
        // DEAD slots (tag -1: reserved early by the Jacobian round and not needed, only their tag is written) and the idle lanes
        // of a list's last tile hold no fit: they do not read the record at all and work on a fixed, valid stand-in instead
        // (an identity R, a zero right-hand side, the start point of a fit), so that no address, table index or LDS offset of
        // this kernel is ever derived from memory nobody wrote.  Nothing such a lane computes is stored (every store below is
        // under `live`).
        int tag = 0, dummy, niter = 1, nfev = 1, ipvt_i = 0x76543210;
        {
            int t;
            unpack2(qb[A_IDX * cap], &t, &dummy);
            if (t == -1) live = false;
            if (live) tag = t;
        }
        QuadLm q;
        double xq[FSQ_NP];
        double llim1 = 0., fnorm = 1., par = 0., delta = 1., xnorm = 1., fnorm1;
        if (live) {
            unpack2(qb[A_ITER * cap], &niter, &nfev);
            unpack2(qb[B_IPVT * cap], &ipvt_i, &dummy);
#pragma unroll
            for (int i = 0; i < FSQ_NP; i++)
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) q.r[i][k] = (k >= i) ? (double)qb[(B_R + rpk(i, k)) * cap] : 0.0;
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) {
                q.qtf[k] = qb[(B_QTF + k) * cap]; q.dg[k] = qb[(A_DIAG + k) * cap]; q.sdiag[k] = qb[(B_SDIAG + k) * cap];
                xq[k] = qb[(A_X + k) * cap];
            }
            llim1 = qb[A_LLIM1 * cap];
            fnorm = qb[A_FNORM * cap]; par = qb[A_PAR * cap]; delta = qb[A_DELTA * cap]; xnorm = qb[A_XNORM * cap];
        } else {
            const double x0[FSQ_NP] = {100., 1000., 2.5, 2.5, 1., 1., 0.};
#pragma unroll
            for (int i = 0; i < FSQ_NP; i++)
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) q.r[i][k] = (k == i) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) { q.qtf[k] = 0.; q.dg[k] = 1.; q.sdiag[k] = 0.; xq[k] = x0[k]; }
        }
        const unsigned ipvt = (unsigned)ipvt_i;
        // (gnorm is only copied through and tested once at the end: it is read where it is needed instead of being
        // carried through lmpar - the kernel is at its register limit)
#define KB_GNORM() kb_late_load(qb[B_GNORM * cap])
#pragma unroll
        for (int k = 0; k < FSQ_NP; k++) myscr[k * 64] = q.dg[k];
#pragma unroll
        for (int j = 0; j < FSQ_NP; j++) q.dgp[j] = myscr[nib_get(ipvt, j) * 64];

        RPH_MARK(1)
        int lm_hist = 0;                        // Newton iterations this lmpar call took in all: the fit's history for its next record
        {
            QuadLmparSt st;
            if (!resume) quadlm_lmpar_begin<ALIASED, 64>(q, myscr, ipvt, delta, par, st);
            else if (live) {
                int it, dm;
                unpack2(qb[C_LMIT * cap], &it, &dm);
                st.par = par; st.parl = qb[C_PARL * cap]; st.paru = qb[C_PARU * cap]; st.fp = qb[C_FP * cap];
                st.iter = it; st.done = false;
            } else { st.par = 0.; st.parl = 0.; st.paru = 0.; st.fp = 0.; st.iter = 10; st.done = true; }
            RPH_MARK(7)
            quadlm_lmpar_run<ALIASED, 64>(q, myscr, ipvt, delta, st, lm_limit);
            par = st.par;
            lm_hist = st.iter;
            {
                // unfinished: park the fit (R / sdiag / par state as they stand) for the next round's launch
                const bool park = live && !st.done;
                const bool to_c1 = (st.iter < lims.lim[2]);            // (C 1 tiles run to lim[2], C 3 tiles to the end: always progress)
                const int at1 = (int)wave_reserve_checked(c, cnt_next + CNT_C1, park && to_c1, cap, G_KB_C1);
                const int at3 = (int)wave_reserve_checked(c, cnt_next + CNT_C3, park && !to_c1, cap, G_KB_C3);
                if (park) {
                    const NtQ qn = ntq(QC_next + (to_c1 ? (long long)at1 : cap - 1 - at3));
                    qn[A_IDX * cap] = pack2(tag, 0);
#pragma unroll
                    for (int k = 0; k < FSQ_NP; k++) {
                        qn[(A_X + k) * cap] = xq[k]; qn[(A_DIAG + k) * cap] = q.dg[k];
                        qn[(B_QTF + k) * cap] = q.qtf[k]; qn[(B_SDIAG + k) * cap] = q.sdiag[k];
                    }
                    qn[A_LLIM1 * cap] = llim1; qn[A_FNORM * cap] = fnorm; qn[A_PAR * cap] = par; qn[A_DELTA * cap] = delta;
                    qn[A_XNORM * cap] = xnorm; qn[A_ITER * cap] = pack2(niter, nfev);
                    qn[B_GNORM * cap] = KB_GNORM(); qn[B_IPVT * cap] = pack2((int)ipvt, 0);
#pragma unroll
                    for (int i = 0; i < FSQ_NP; i++)
#pragma unroll
                        for (int k = i; k < FSQ_NP; k++) qn[(B_R + rpk(i, k)) * cap] = q.r[i][k];
                    qn[C_PARL * cap] = st.parl; qn[C_PARU * cap] = st.paru; qn[C_FP * cap] = st.fp;
                    qn[C_LMIT * cap] = pack2(st.iter, 0);
                }
                live = live && !park;           // a parked lane idles through the rest of this pass
            }
        }
        RPH_MARK(2)
        double wa1[FSQ_NP], wa2[FSQ_NP];
        bool lpeg[FSQ_NP], upeg[FSQ_NP];
        int nlpeg = 0, nupeg = 0;
#pragma unroll
        for (int k = 0; k < FSQ_NP; k++) {
            wa1[k] = -q.xp[k];
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
        double pnorm = 0.0;
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
            double t = q.dg[k] * wa1[k];
            pnorm = fsq_fma(t, t, pnorm);
            myscr[(14 + k) * 64] = wa1[k];
        }
        pnorm = fsq_sqrt(pnorm);
        if (niter == 1) delta = np_min2(delta, pnorm);
        RPH_MARK(3)
        // trial evaluation (mpfit.py:1245): E per pixel into the LDS column, then fnorm1 = enorm(data - (p0 + p1 * E)) with
        // the residuals formed on the fly in numpy.dot's order (dot25: four strided sums of four, then an fma tail), so no
        // 25-element residual array lives in registers
        {
            asm volatile("" ::: "memory");          // the pixels are fetched here, not carried through lmpar
            bool redo = kb_trial_gauss<true>(wa2, myscr);
#ifdef FSQ_EXPERIMENT_TRIAL_TWICE       // marginal cost of the trial evaluation: do it again (same results)
            asm volatile("" ::: "memory");
            redo = kb_trial_gauss<true>(wa2, myscr) || redo;
            asm volatile("" ::: "memory");
#endif
            if (c.force_slow_mod > 0 && (tag_slot(c, tag) % c.force_slow_mod) == 0) redo = true;
            if (__ballot(redo)) {                   // (never on image data: operands far outside the guarded ranges)
                if (redo) kb_trial_gauss<false>(wa2, myscr);
            }
            const uint16_t* src = c.roi + (size_t)tag_slot(c, tag) * 32;
            unsigned w[P32 ? 28 : 16];
            if (P32) {
                const uint32_t* s32 = (const uint32_t*)c.roi + (size_t)tag_slot(c, tag) * 32;
#pragma unroll
                for (int g = 0; g < 7; g++) { const uint4 q4 = nt_ld4(s32 + 4 * g); w[4 * g] = q4.x; w[4 * g + 1] = q4.y; w[4 * g + 2] = q4.z; w[4 * g + 3] = q4.w; }
            } else {
                const uint4 a = nt_ld4(src), b = nt_ld4(src + 8), e = nt_ld4(src + 16), f = nt_ld4(src + 24);
                const unsigned w16[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, e.x, e.y, e.z, e.w, f.x, f.y, f.z, f.w};
#pragma unroll
                for (int g = 0; g < 16; g++) w[g] = w16[g];
            }
            double S[4], d = 0.0;
#pragma unroll
            for (int i = 0; i < FSQ_NPIX; i++) {
                const double pix = P32 ? (double)w[i] : (double)((w[i >> 1] >> (16 * (i & 1))) & 0xffffu);
                const double r = pix - (wa2[0] + wa2[1] * myscr[kb_res_slot(i) * 64]);
                if (i < 4) S[i] = r * r;
                else if (i < 16) S[i & 3] = S[i & 3] + r * r;
                else {
                    if (i == 16) d = (S[0] + S[2]) + (S[1] + S[3]);
                    d = fsq_fma(r, r, d);
                }
            }
            fnorm1 = fsq_sqrt(d);
        }
        nfev++;
        RPH_MARK(4)
        double actred = -1.;
        if ((0.1 * fnorm1) < fnorm) actred = -fsq_pow2(fnorm1 / fnorm) + 1.;
        double wa3[FSQ_NP];
#pragma unroll
        for (int k = 0; k < FSQ_NP; k++) wa3[k] = 0.;
#pragma unroll
        for (int j = 0; j < FSQ_NP; j++) {
            wa3[j] = 0;
            double wj = myscr[(14 + nib_get(ipvt, j)) * 64];
#pragma unroll
            for (int i = 0; i < FSQ_NP; i++)
                if (i <= j) wa3[i] = wa3[i] + q.r[i][j] * wj;
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
        const bool accepted = (ratio >= 0.0001);
        if (accepted) {
            double xs = 0.0;
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) {
                xq[k] = wa2[k];
                double t = q.dg[k] * wa2[k];
                xs = fsq_fma(t, t, xs);
            }
            xnorm = fsq_sqrt(xs);
            fnorm = fnorm1;
            niter = niter + 1;
        }
        int status = 0;
        bool c1 = (__builtin_fabs(actred) <= 1e-10) && (prered <= 1e-10) && (0.5 * ratio <= 1);
        if (c1) status = 1;
        if (delta <= 1e-10 * xnorm) status = 2;
        if (c1 && (status == 2)) status = 3;
        if (status == 0) {
            if (niter >= 200) status = 5;
            if ((__builtin_fabs(actred) <= FSQ_MACHEP) && (prered <= FSQ_MACHEP) && (0.5 * ratio <= 1)) status = 6;
            if (delta <= FSQ_MACHEP * xnorm) status = 7;
            if (KB_GNORM() <= FSQ_MACHEP) status = 8;
        }
        if (status == 0 && !accepted) {
            bool fin = __builtin_isfinite(ratio);
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++)
                fin = fin && __builtin_isfinite(wa1[k]) && __builtin_isfinite(wa2[k]) && __builtin_isfinite(xq[k]);
            if (!fin) status = -16;
        }
        RPH_MARK(5)
        if (live && status != 0) {
            FitOut o;
#pragma unroll
            for (int k = 0; k < FSQ_NP; k++) o.x[k] = xq[k];
            o.status = status; o.niter = niter; o.nfev = nfev; o.pad = 0;
            c.out[tag_slot(c, tag)] = o;
        }
        wave_mark_done(c.done, live && status != 0, tag_ticket(c, tag));
        {
            // accepted -> a new Jacobian (queue A); rejected -> another pass with the same, mutated R (queue B)
            const bool toA = live && status == 0 && accepted, toB = live && status == 0 && !accepted;
            const bool hi = lm_hist > 1;
#ifdef FSQ_EXPERIMENT_EXTRA_ATOMICS_KB
            {
                int sink = 0;
                for (int x = 0; x < FSQ_EXPERIMENT_EXTRA_ATOMICS_KB; x++) sink += wave_reserve(cnt_next + 5 + (x & 1), live);
                if (sink == 0x7fffffff) atomicMax(c.err, 99);
            }
#endif
            const int atA = (int)wave_reserve_checked(c, cnt_next + CNT_A, toA, cap, G_KB_A);
            const int atBl = (int)wave_reserve_checked(c, cnt_next + CNT_BLO, toB && !hi, cap, G_KB_BLO);
            const int atBh = (int)wave_reserve_checked(c, cnt_next + CNT_BHI, toB && hi, cap, G_KB_BHI);
            if (toA || toB) {
                const NtQ qn = ntq(toA ? (QA_next + atA) : (QB_next + (hi ? cap - 1 - atBh : (long long)atBl)));
                qn[A_IDX * cap] = pack2(tag, lm_hist);
#pragma unroll
                for (int k = 0; k < FSQ_NP; k++) { qn[(A_X + k) * cap] = xq[k]; qn[(A_DIAG + k) * cap] = q.dg[k]; }
                qn[A_LLIM1 * cap] = llim1; qn[A_FNORM * cap] = fnorm; qn[A_PAR * cap] = par; qn[A_DELTA * cap] = delta;
                qn[A_XNORM * cap] = xnorm; qn[A_ITER * cap] = pack2(niter, nfev);
                if (toA) {      // E at the accepted point travels with the record (coalesced: the lanes' positions are consecutive)
#pragma unroll
                    for (int i = 0; i < FSQ_NPIX; i++) qn[(A_E + i) * cap] = myscr[kb_res_slot(i) * 64];
                }
                if (toB) {
                    qn[B_GNORM * cap] = KB_GNORM(); qn[B_IPVT * cap] = pack2((int)ipvt, 0);
#pragma unroll
                    for (int k = 0; k < FSQ_NP; k++) { qn[(B_QTF + k) * cap] = q.qtf[k]; qn[(B_SDIAG + k) * cap] = q.sdiag[k]; }
#undef KB_GNORM
#pragma unroll
                    for (int i = 0; i < FSQ_NP; i++)
#pragma unroll
                        for (int k = i; k < FSQ_NP; k++) qn[(B_R + rpk(i, k)) * cap] = q.r[i][k];
                }
            }
        }
        RPH_MARK(6)
    }
    RPH_FLUSH(0)
}

// ---------------------------------------------------------------------------------------------------
// fit-quality metrics and the output row (pflib.py:461-475)
__global__ void __launch_bounds__(64) kfinish(Ctx c, BatchArgs b, FsqRow* __restrict__ rows)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n) return;
    const long long slot = b.base + i;
    const FitOut S = c.out[slot];
    double data[FSQ_NPIX], p[FSQ_NP];
    roi_compact(c, slot, data);
#pragma unroll
    for (int k = 0; k < FSQ_NP; k++) p[k] = S.x[k];
    const double vmax = c.stat[slot].vmax, vmean = c.stat[slot].vmean;
    int h = 2, w = 2, field = 0;
    if (b.from_image) { field = b.cand[3 * i]; h = b.cand[3 * i + 1]; w = b.cand[3 * i + 2]; }
    double fit[FSQ_NPIX];
    fsq_model(p, fit);
    double num = 0.0, den = 0.0, rm = 0.0;
    for (int k = 0; k < FSQ_NPIX; k++) { double d = data[k] - fit[k]; num += d * d; }
    for (int k = 0; k < FSQ_NPIX; k++) { double d = data[k] - vmean; den += d * d; }
    for (int k = 0; k < FSQ_NPIX; k++) rm += fsq_pow2(data[k] - fit[k]);
    FsqRow r;
    r.h0 = p[2] + h - 2.5;                                              // pflib.py:461
    r.w0 = p[3] + w - 2.5;
    r.H = p[0]; r.A = p[1]; r.sigma_h = p[4]; r.sigma_w = p[5]; r.theta = p[6];
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
    r.p2 = p[2]; r.p3 = p[3];
    r.h = h; r.w = w; r.field = field;
    const int status = S.status;
    r.status = status; r.niter = S.niter; r.nfev = S.nfev + (status > 0 ? 1 : 0);   // mpfit's final call (mpfit.py:1353)
    r.key_h = -1; r.key_w = -1;
    rows[i] = r;
}

}  // namespace

// ---- self-test hook: fsq_div_by against the compiler's `/` on caller-supplied operand pairs ------------------
namespace {
__global__ void kdivcheck(const double* __restrict__ num, const double* __restrict__ den, long long n, unsigned long long* bad)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = num[i], d = den[i];
    const FsqDivisor k = fsq_divisor(d);
    const double q0 = a / d, q1 = fsq_div_by(a, k);
    if (fsq_bits(q0) != fsq_bits(q1)) atomicAdd(bad, 1ull);
}
std::atomic<long long> g_last_slow{0};

__global__ void ksqcheck(const double* __restrict__ t, long long n, unsigned long long* bad)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = t[i], sq = x * x, lo = fsq_fma(x, x, -sq);
    if (!fsq_square_is_pow2(x, sq, lo)) { atomicAdd(bad + 1, 1ull); return; }
    if (fsq_bits(fsq_pow2(x)) != fsq_bits(sq)) atomicAdd(bad, 1ull);
}

__global__ void krotcheck(const double* __restrict__ t, long long n, unsigned long long* bad)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = .25 + .25 * t[i] * t[i];
    const double a = 0.5 / fsq_sqrt(x), b = fsq_half_over_sqrt_q(x);
    if (fsq_bits(a) != fsq_bits(b) && !(a != a && b != b)) atomicAdd(bad, 1ull);
}
}  // namespace

namespace {
__global__ void kexpcheck(const double* __restrict__ x, long long n, unsigned long long* bad)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool flag = false;
    const double a = fsq_exp(x[i]), b = fsq_exp_bf(x[i], &flag);
    const bool expect_flag = !(__builtin_fabs(x[i]) < 512.0);             // (NaN included)
    bool wrong = (flag != expect_flag);
    if (!expect_flag) wrong = wrong || (fsq_bits(a) != fsq_bits(b));
    if (wrong) atomicAdd(bad, 1ull);
}
}  // namespace

extern "C" int fsq_selftest_exp(const double* d_x, int64_t n, int64_t* mismatches, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!d_x || !mismatches || n < 0) return FSQ_EINVAL;
    unsigned long long* d_bad = nullptr;
    FSQ_HIP_CHECK(hipMalloc((void**)&d_bad, 8));
    FSQ_HIP_CHECK(hipMemsetAsync(d_bad, 0, 8, s));
    if (n > 0) hipLaunchKernelGGL(kexpcheck, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_x, (long long)n, d_bad);
    unsigned long long h = 0;
    FSQ_HIP_CHECK(hipMemcpyAsync(&h, d_bad, 8, hipMemcpyDeviceToHost, s));
    FSQ_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(d_bad);
    *mismatches = (int64_t)h;
    return FSQ_OK;
}

extern "C" int fsq_selftest_square(const double* d_t, int64_t n, int64_t* mismatches, int64_t* undecided, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!d_t || !mismatches || !undecided || n < 0) return FSQ_EINVAL;
    unsigned long long* d_bad = nullptr;
    FSQ_HIP_CHECK(hipMalloc((void**)&d_bad, 16));
    FSQ_HIP_CHECK(hipMemsetAsync(d_bad, 0, 16, s));
    if (n > 0) hipLaunchKernelGGL(ksqcheck, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_t, (long long)n, d_bad);
    unsigned long long h[2] = {0, 0};
    FSQ_HIP_CHECK(hipMemcpyAsync(h, d_bad, 16, hipMemcpyDeviceToHost, s));
    FSQ_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(d_bad);
    *mismatches = (int64_t)h[0];
    *undecided = (int64_t)h[1];
    return FSQ_OK;
}

extern "C" int fsq_selftest_rotation(const double* d_t, int64_t n, int64_t* mismatches, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!d_t || !mismatches || n < 0) return FSQ_EINVAL;
    unsigned long long* d_bad = nullptr;
    FSQ_HIP_CHECK(hipMalloc((void**)&d_bad, 8));
    FSQ_HIP_CHECK(hipMemsetAsync(d_bad, 0, 8, s));
    if (n > 0) hipLaunchKernelGGL(krotcheck, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_t, (long long)n, d_bad);
    unsigned long long h = 0;
    FSQ_HIP_CHECK(hipMemcpyAsync(&h, d_bad, 8, hipMemcpyDeviceToHost, s));
    FSQ_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(d_bad);
    *mismatches = (int64_t)h;
    return FSQ_OK;
}

extern "C" int fsq_selftest_division(const double* d_num, const double* d_den, int64_t n, int64_t* mismatches, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!d_num || !d_den || !mismatches || n < 0) return FSQ_EINVAL;
    unsigned long long* d_bad = nullptr;
    FSQ_HIP_CHECK(hipMalloc((void**)&d_bad, 8));
    FSQ_HIP_CHECK(hipMemsetAsync(d_bad, 0, 8, s));
    if (n > 0) hipLaunchKernelGGL(kdivcheck, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_num, d_den, (long long)n, d_bad);
    unsigned long long h = 0;
    FSQ_HIP_CHECK(hipMemcpyAsync(&h, d_bad, 8, hipMemcpyDeviceToHost, s));
    FSQ_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(d_bad);
    *mismatches = (int64_t)h;
    return FSQ_OK;
}

extern "C" int64_t fsq_fit_last_slow_count(void) { return g_last_slow.load(); }
#ifdef FSQ_DEBUG_HZ
extern "C" int fsq_debug_hz(unsigned long long* out32, int reset)
{
    FSQ_HIP_CHECK(hipDeviceSynchronize());
    FSQ_HIP_CHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_hz), sizeof(unsigned long long) * 32));
    if (reset) { unsigned long long z[32] = {0}; FSQ_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_hz), z, sizeof(z))); }
    return FSQ_OK;
}
#endif

#ifdef FSQ_PHASE_PROFILE
extern "C" int fsq_debug_rphase(unsigned long long* out32, int reset)
{
    unsigned long long* p = nullptr;
    if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_rphase)) != hipSuccess) return -1;
    if (hipMemcpy(out32, p, 32 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (reset) (void)hipMemset(p, 0, 32 * 8);
    return 0;
}
#endif

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- host side: the rounds engine ---------------------------------------------------------------------------------
// One engine = a pool of by-candidate slots + the work queues + the round loop.  Batches of candidates are SUBMITTED to
// it (kinit appends their fresh records to the current Jacobian queue) and every round advances all fits in flight,
// whatever batch they belong to; a batch is FINISHED (kfinish writes its rows) when its done counter reaches its size.
// fsq_fit_candidates / fsq_fit_rois run one batch through a temporary engine laid out in the caller's workspace;
// FsqFitQueue (include/fsq.h) keeps an engine alive across batches, so that the long latency-bound tail of one batch
// (a fit may need 200 sequential iterations) rides along in the full launches of the batches submitted after it.
namespace {
enum { CTL_SLOW_TOTAL = 2 * CNT_SET, CTL_SLOW_CNT = 2 * CNT_SET + 1, CTL_F32_NEXT = 2 * CNT_SET + 2, CTL_ERR = 2 * CNT_SET + 3, CTL_DONE = 2 * CNT_SET + 8, CTL_INTS = CTL_DONE + FSQ_MAX_TICKETS };

struct RoundsCfg {
    int lm_first = FSQ_LMPAR_FIRST, lm_lo = FSQ_LMPAR_LO, sync_mask = 3, force_slow_mod = 0, force_redo = 0, no_wave_prio = 0, trace = 0;
    int ka_lanes = FSQ_KA_LANES_DEFAULT;               // lanes per fit in the Jacobian round (4 or 8)
    int max_rounds = 0, ka_lds_pad = 0, kb_lds_pad = 0;      // (debug: extra dynamic LDS per block = fewer waves per CU)
    long long two_pass_min = 524288, hiprio_below = 200000;
};
RoundsCfg read_cfg()
{
    RoundsCfg g;
    const char* e;
    if ((e = getenv("FSQ_DEBUG_FORCE_SLOW")) != nullptr) g.force_slow_mod = atoi(e);
    if ((e = getenv("FSQ_DEBUG_FORCE_NORM_RECOMPUTE")) != nullptr) g.force_redo = atoi(e) ? 1 : 0;
    if ((e = getenv("FSQ_LMPAR_FIRST_ITERS")) != nullptr && atoi(e) >= 1 && atoi(e) <= 10) g.lm_first = atoi(e);
    if ((e = getenv("FSQ_LMPAR_LO_ITERS")) != nullptr && atoi(e) >= 1 && atoi(e) <= 10) g.lm_lo = atoi(e);
    if ((e = getenv("FSQ_TWO_PASS_MIN")) != nullptr) g.two_pass_min = atoll(e);
    if ((e = getenv("FSQ_HIPRIO_BELOW")) != nullptr) g.hiprio_below = atoll(e);
    if ((e = getenv("FSQ_SYNC_EVERY")) != nullptr && atoi(e) >= 1) g.sync_mask = atoi(e) - 1;     // power of two
    if (getenv("FSQ_NO_WAVE_PRIO")) g.no_wave_prio = 1;
    if (getenv("FSQ_DEBUG_TRACE")) g.trace = 1;
    if ((e = getenv("FSQ_DEBUG_MAX_ROUNDS")) != nullptr) g.max_rounds = atoi(e);
    if ((e = getenv("FSQ_KA_LANES")) != nullptr && (atoi(e) == 4 || atoi(e) == 8)) g.ka_lanes = atoi(e);
    if ((e = getenv("FSQ_DEBUG_KA_LDS_PAD")) != nullptr) g.ka_lds_pad = atoi(e);
    if ((e = getenv("FSQ_DEBUG_KB_LDS_PAD")) != nullptr) g.kb_lds_pad = atoi(e);
    return g;
}

// High-priority streams for the late rounds of a stand-alone batch.  Once few fits are left a round is a handful of
// small kernels whose latency is the whole cost; when another stream (another lane of engine.LanePipeline) is busy with
// the large early rounds of its own batch, the dispatcher would queue those small kernels behind whole large ones.  From
// the first host look that finds fewer than FSQ_HIPRIO_BELOW live fits the rounds therefore continue on a stream of the
// highest priority (the switch happens right after a stream synchronisation, so no event is needed).
struct HiStream { hipStream_t s; int dev; bool busy; };
std::mutex g_hi_mu;
std::vector<HiStream> g_hi;
hipStream_t hi_acquire()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_hi_mu);
    for (auto& h : g_hi)
        if (!h.busy && h.dev == dev) { h.busy = true; return h.s; }
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return nullptr;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest) != hipSuccess) return nullptr;
    g_hi.push_back({st, dev, true});
    return st;
}
void hi_release(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_hi_mu);
    for (auto& h : g_hi)
        if (h.s == st) h.busy = false;
}

enum { FSQ_TICKET_BITS = 5 };
static_assert((1 << FSQ_TICKET_BITS) == FSQ_MAX_TICKETS, "ticket bits");
enum { T_FREE = 0, T_FLIGHT = 1, T_FINISHED = 2 };
struct Batch { BatchArgs a; FsqRow* rows; int state; hipEvent_t ev; };

// Queue capacity in positions: field f of the record at position i lives at q[f * cap + i], and a wave reads / writes 64
// consecutive positions of a field at a time - with cap a multiple of 256 every such 512-byte piece is aligned, whatever
// capacity the caller asked for (an odd cap measured 6 % slower: every access straddled one more cache line).
size_t cap_round(size_t q) { return (q + 255) & ~(size_t)255; }

size_t layout_bytes(size_t pool, size_t qcap)
{
    qcap = cap_round(qcap);
    size_t b = 4096;
    b += al256(pool * 128) + al256(pool * sizeof(FitOut)) + al256(pool * sizeof(FitStat));      // (ROI copies: room for 32-bit pixels)
    b += 2 * al256(qcap * A_LEN * 8) + 2 * al256(qcap * B_LEN * 8) + 2 * al256(qcap * C_LEN * 8) + al256(qcap * A_LEN * 8);
    return b;
}
}  // namespace

struct FsqFitQueue {
    Ctx c;
    size_t pool = 0, qcap = 0;
    int* ctl = nullptr;
    double *QA[2], *QB[2], *QC[2], *SQ = nullptr;
    int *cset[2], *cSlow = nullptr;
    RoundsCfg cfg;
    bool ref = true, f32 = false, single_call = false;
    int cus = 256;
    long long round = 0;                                   // rounds run so far; set (round & 1) is consumed next
    long long boundA = 0, alive = 0, slow_pending = 0;     // host-side upper bounds: queue A of the current set, all fits in flight
    long long head = 0;                                    // ring allocator over the pool slots
    Batch b[FSQ_MAX_TICKETS];
    hipStream_t s = nullptr, s_finish = nullptr, hi = nullptr;
    int h_ctl[CTL_INTS];
    void* owned_ws = nullptr;

    int init(void* d_ws, int64_t ws_bytes, size_t pool_, size_t qcap_, int mode, hipStream_t stream, bool single)
    {
        if (!d_ws || (size_t)ws_bytes < layout_bytes(pool_, qcap_)) return FSQ_ENOMEM;
        qcap_ = cap_round(qcap_);
        if (pool_ > 2000000000ull || qcap_ > 2000000000ull) return FSQ_ENOTIMPL;
        pool = pool_; qcap = qcap_; s = s_finish = stream; single_call = single;
        ref = ((mode & 0xff) == FSQ_MODE_REF);
        f32 = ((mode & 0xff) == FSQ_MODE_TEXTBOOK_F32);
        cfg = read_cfg();
        unsigned char* ws = (unsigned char*)d_ws;
        ctl = (int*)ws;     // two sets of queue counters (CNT_*), slow total, slow queue, done counters per ticket
        size_t o = 4096;
        c.cap = (long long)qcap;
        c.roi = (uint16_t*)(ws + o); o += al256(pool * 128);
        c.pix32 = (mode & FSQ_PIXELS_U32_FLAG) ? 1 : 0;
        c.out = (FitOut*)(ws + o); o += al256(pool * sizeof(FitOut));
        c.stat = (FitStat*)(ws + o); o += al256(pool * sizeof(FitStat));
        QA[0] = (double*)(ws + o); o += al256(qcap * A_LEN * 8);
        QA[1] = (double*)(ws + o); o += al256(qcap * A_LEN * 8);
        QB[0] = (double*)(ws + o); o += al256(qcap * B_LEN * 8);
        QB[1] = (double*)(ws + o); o += al256(qcap * B_LEN * 8);
        QC[0] = (double*)(ws + o); o += al256(qcap * C_LEN * 8);
        QC[1] = (double*)(ws + o); o += al256(qcap * C_LEN * 8);
        SQ = (double*)(ws + o); o += al256(qcap * A_LEN * 8);
        cset[0] = ctl; cset[1] = ctl + CNT_SET;
        cSlow = ctl + CTL_SLOW_CNT;
        c.slow_total = ctl + CTL_SLOW_TOTAL; c.done = ctl + CTL_DONE; c.err = ctl + CTL_ERR; c.pool = (long long)pool;
        c.tshift = single ? 31 : 32 - FSQ_TICKET_BITS;
        if (!single && pool_ >= (1ull << c.tshift)) return FSQ_ENOTIMPL;
        c.wave_prio = 0; c.force_redo = cfg.force_redo; c.force_slow_mod = cfg.force_slow_mod;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        for (auto& t : b) { t.state = T_FREE; t.ev = nullptr; t.rows = nullptr; }
        FSQ_HIP_CHECK(hipMemsetAsync(ctl, 0, CTL_INTS * sizeof(int), s));
        return FSQ_OK;
    }

    // contiguous range of n pool slots that no batch in flight owns (ring order); -1 if there is none right now
    long long alloc_slots(long long n)
    {
        if ((size_t)n > pool) return -1;
        long long at = head;
        for (int attempt = 0; attempt < 2; attempt++) {
            if ((size_t)(at + n) > pool) at = 0;
            bool clash = false;
            for (const auto& t : b)
                if (t.state == T_FLIGHT && at < t.a.base + t.a.n && t.a.base < at + n) { clash = true; break; }
            if (!clash) { head = at + n; return at; }
            if (at == 0) break;
            at = 0;
        }
        return -1;
    }

    int submit(const uint16_t* src, int pix_fmt, int H, int W, const int32_t* cand, long long n, bool from_image, FsqRow* rows, int* ticket)
    {
        if (n < 0 || (n > 0 && (!src || !rows || (from_image && !cand)))) return FSQ_EINVAL;
        if (pix_fmt != FSQ_PIXELS_U16 && pix_fmt != FSQ_PIXELS_F16 && pix_fmt != FSQ_PIXELS_U32) return FSQ_EINVAL;
        // 32-bit pixels: an engine created for them (FSQ_PIXELS_U32_FLAG in its mode) takes nothing else, and the other way round
        // (the compact ROI copies of all batches in flight share one word size); not in the single-precision mode, not with 8 lanes
        if ((pix_fmt == FSQ_PIXELS_U32) != (c.pix32 != 0)) return FSQ_ENOTIMPL;
        if (c.pix32 && (f32 || cfg.ka_lanes == 8)) return FSQ_ENOTIMPL;
        int t = -1;
        for (int k = 0; k < FSQ_MAX_TICKETS; k++)
            if (b[k].state == T_FREE) { t = k; break; }
        if (t < 0) return FSQ_EAGAIN;
        if (!f32 && (size_t)(alive + n) > qcap) return FSQ_EAGAIN;      // (the single-precision solver does not use the queues)
        const long long base = n > 0 ? alloc_slots(n) : 0;
        if (base < 0) return FSQ_EAGAIN;
        Batch& B = b[t];
        B.a.src = src; B.a.cand = cand; B.a.H = H; B.a.W = W; B.a.n = n; B.a.from_image = from_image ? 1 : 0; B.a.pix_fmt = pix_fmt;
        B.a.base = base; B.a.ticket = t; B.a.no_queue = f32 ? 1 : 0; B.rows = rows;
        if (!single_call && !B.ev) FSQ_HIP_CHECK(hipEventCreateWithFlags(&B.ev, hipEventDisableTiming));
        if (n == 0) {
            B.state = T_FINISHED;
            if (B.ev) FSQ_HIP_CHECK(hipEventRecord(B.ev, s));
        } else if (f32) {
            // single precision: no rounds - the whole batch is fitted by one persistent launch (fsq_fit_f32.h), in stream order
            FSQ_HIP_CHECK(hipMemsetAsync(ctl + CTL_F32_NEXT, 0, sizeof(int), s));
            hipLaunchKernelGGL(kinit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c, B.a, QA[0], cset[0] + CNT_A);
            const long long waves = std::min<long long>((n + 63) / 64, (long long)cus * 12);          // 3 waves per SIMD (164 VGPRs)
            hipLaunchKernelGGL(kfit_f32, dim3((unsigned)waves), dim3(64), 0, s, c, B.a, ctl + CTL_F32_NEXT);
            hipLaunchKernelGGL(kfinish, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, c, B.a, rows);
            if (B.ev) FSQ_HIP_CHECK(hipEventRecord(B.ev, s));
            B.state = T_FINISHED;
        } else {
            const int cur = (int)(round & 1);
            FSQ_HIP_CHECK(hipMemsetAsync(c.done + t, 0, sizeof(int), s));
            hipLaunchKernelGGL(kinit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c, B.a, QA[cur], cset[cur] + CNT_A);
            B.state = T_FLIGHT;
            boundA += n; alive += n;
        }
        if (ticket) *ticket = t;
        return FSQ_OK;
    }

    // One round: Jacobian round over queue A of set cur (+ the slow queue), step round over queues B and C; results in set nxt.
    int one_round()
    {
        const int cur = (int)(round & 1), nxt = cur ^ 1;
        // Grid sizing: one block per tile (16 or 8 fits in kA, 64 in kB), so blocks retire continuously.  The hardware
        // dispatcher then balances the very uneven tile times, and small kernels of ANOTHER stream find free CU slots
        // between them.
        const int G = 64 / cfg.ka_lanes;
        const long long gA = (boundA + G - 1) / G;
        if (slow_pending > 0) {
            // fits that left the guarded operand ranges in EARLIER rounds: the plain-division build takes them from the slow
            // queue (whatever it holds by now - it may have grown since the host looked) and appends their queue-B records to
            // this round's.  It runs BEFORE this round's fast kernel: a fit the fast kernel sends to the slow queue has already
            // reserved a (dead) queue-B slot in this round, and worked off in the same round it would take a second one - the
            // step round's grid is sized for one slot per live fit, and tiles beyond it would never run (a lost fit, a batch
            // that never finishes; found by the round-3 fuzz on noise fields, where hundreds of fits take this path at once).
            if (c.pix32) hipLaunchKernelGGL((kA_jacobian<false, 4, true>), dim3((unsigned)std::min<long long>((alive + 15) / 16, (long long)cus * 8)), dim3(64), 0, s, c, SQ, cSlow, QB[cur], cset[cur], SQ, cSlow, cset[nxt]);
            else hipLaunchKernelGGL((kA_jacobian<false, 4>), dim3((unsigned)std::min<long long>((alive + 15) / 16, (long long)cus * 8)), dim3(64), 0, s, c, SQ, cSlow, QB[cur], cset[cur], SQ, cSlow, cset[nxt]);
            FSQ_HIP_CHECK(hipMemsetAsync(cSlow, 0, sizeof(int), s));
            slow_pending = 0;
        }
        if (gA > 0 && c.pix32)
            hipLaunchKernelGGL((kA_jacobian<true, 4, true>), dim3((unsigned)gA), dim3(64), cfg.ka_lds_pad, s, c, QA[cur], cset[cur] + CNT_A, QB[cur], cset[cur], SQ, cSlow, cset[nxt]);
        else if (gA > 0 && cfg.ka_lanes == 8)            // (kA also zeroes the counters of set nxt)
            hipLaunchKernelGGL((kA_jacobian<true, 8>), dim3((unsigned)gA), dim3(64), cfg.ka_lds_pad, s, c, QA[cur], cset[cur] + CNT_A, QB[cur], cset[cur], SQ, cSlow, cset[nxt]);
        else if (gA > 0)
            hipLaunchKernelGGL((kA_jacobian<true, 4>), dim3((unsigned)gA), dim3(64), cfg.ka_lds_pad, s, c, QA[cur], cset[cur] + CNT_A, QB[cur], cset[cur], SQ, cSlow, cset[nxt]);
        else
            FSQ_HIP_CHECK(hipMemsetAsync(cset[nxt], 0, CNT_SET * sizeof(int), s));
        // every fit in flight is in one of the four input lists of the step round by now (or terminated, or in the slow queue)
        const long long gB = (alive + 63) / 64 + 4;
        {
            // With few fits left a round is pure launch + wave latency: lmpar then runs to the end wherever a fit is met
            // (nothing is parked, no fit waits for the next round).
            const bool staged = alive > cfg.two_pass_min && cfg.lm_first < 10;
            KbLimits lims;
            lims.lim[0] = staged ? std::min(cfg.lm_lo, cfg.lm_first) : 10; lims.lim[1] = staged ? cfg.lm_first : 10; lims.lim[2] = staged ? cfg.lm_first : 10; lims.lim[3] = 10;
            if (c.pix32 && ref) hipLaunchKernelGGL((kB_step<true, true>), dim3((unsigned)gB), dim3(64), 0, s, c, QB[cur], QC[cur], cset[cur], QA[nxt], QB[nxt], QC[nxt], cset[nxt], lims);
            else if (c.pix32) hipLaunchKernelGGL((kB_step<false, true>), dim3((unsigned)gB), dim3(64), 0, s, c, QB[cur], QC[cur], cset[cur], QA[nxt], QB[nxt], QC[nxt], cset[nxt], lims);
            else if (ref) hipLaunchKernelGGL((kB_step<true>), dim3((unsigned)gB), dim3(64), cfg.kb_lds_pad, s, c, QB[cur], QC[cur], cset[cur], QA[nxt], QB[nxt], QC[nxt], cset[nxt], lims);
            else hipLaunchKernelGGL((kB_step<false>), dim3((unsigned)gB), dim3(64), 0, s, c, QB[cur], QC[cur], cset[cur], QA[nxt], QB[nxt], QC[nxt], cset[nxt], lims);
        }
        boundA = alive;                             // every fit of this round ends in a list of set nxt or is done
        round++;
        return FSQ_OK;
    }

    // Read the counters (synchronises the stream), refresh the bounds, finish the batches that are complete.
    int look(int* newly_finished)
    {
        FSQ_HIP_CHECK(hipMemcpyAsync(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, s));
        FSQ_HIP_CHECK(hipStreamSynchronize(s));
        const int cur = (int)(round & 1);
        const int* hc = h_ctl + CNT_SET * cur;
        boundA = hc[CNT_A];
        slow_pending = h_ctl[CTL_SLOW_CNT];
        alive = boundA + hc[CNT_BLO] + hc[CNT_BHI] + hc[CNT_C1] + hc[CNT_C3] + slow_pending;
        g_last_slow.store(h_ctl[CTL_SLOW_TOTAL]);
        if (h_ctl[CTL_ERR] != 0) {          // a kernel was about to write outside a queue / the pool (fsq_guard): an engine bug, say which
            fprintf(stderr, "fsq: fit queue invariant %d broken (cap %lld, pool %lld, host bounds: A %lld alive %lld; counters A %d B %d+%d C %d+%d slow %d)\n",
                    h_ctl[CTL_ERR], (long long)qcap, (long long)pool, boundA, alive, hc[CNT_A], hc[CNT_BLO], hc[CNT_BHI], hc[CNT_C1], hc[CNT_C3], h_ctl[CTL_SLOW_CNT]);
            return FSQ_EINTERNAL;
        }
        if (cfg.trace) fprintf(stderr, "round %lld: A=%lld B=%d+%d C=%d+%d slow=%lld total_slow=%d\n", round - 1, boundA, hc[CNT_BLO], hc[CNT_BHI], hc[CNT_C1], hc[CNT_C3], slow_pending, h_ctl[CTL_SLOW_TOTAL]);
        int fin = 0;
        for (auto& t : b) {
            if (t.state != T_FLIGHT || h_ctl[CTL_DONE + t.a.ticket] < t.a.n) continue;
            // (s_finish: a stand-alone call hands its rows over on the caller's stream, idle and ordered here)
            hipLaunchKernelGGL(kfinish, dim3((unsigned)((t.a.n + 63) / 64)), dim3(64), 0, s_finish, c, t.a, t.rows);
            if (t.ev) FSQ_HIP_CHECK(hipEventRecord(t.ev, s_finish));
            t.state = T_FINISHED;
            fin++;
        }
        if (newly_finished) *newly_finished = fin;
        if (single_call && !hi && alive > 0 && alive < cfg.hiprio_below && (long long)pool >= 4 * cfg.hiprio_below) {
            hi = hi_acquire();                  // (small batches gain nothing)
            if (hi) s = hi;                     // the user's stream is idle here: plain hand-over
            c.wave_prio = cfg.no_wave_prio ? 0 : 1;
        }
        return FSQ_OK;
    }

    // Run rounds until a batch finishes, nothing is left, fewer than `alive_below` fits are alive, or `max_rounds` ran.
    int advance(long long max_rounds, long long alive_below, long long* alive_out, int* finished_out)
    {
        int fin_total = 0;
        long long ran = 0;
        while (alive > 0) {
            int rc = one_round();
            if (rc != FSQ_OK) return rc;
            ran++;
            const bool stop = (max_rounds > 0 && ran >= max_rounds);
            if (stop || (round & cfg.sync_mask) == 0) {      // (rounds on empty queues cost a few empty launches)
                int fin = 0;
                rc = look(&fin);
                if (rc != FSQ_OK) return rc;
                fin_total += fin;
                if (stop || fin_total > 0 || alive < alive_below) break;
            }
            if (round > 100000000ll) return FSQ_EHIP;
        }
        FSQ_HIP_CHECK(hipGetLastError());
        if (alive == 0 && fin_total == 0) {
            // nothing is alive, so every batch in flight must be complete; one that is not has lost a fit (an engine bug): say so
            // instead of letting the caller wait for it for ever
            bool flight = false;
            for (const auto& t : b) flight = flight || (t.state == T_FLIGHT);
            if (flight) {
                int fin = 0;
                const int rc = look(&fin);
                if (rc != FSQ_OK) return rc;
                fin_total += fin;
                for (const auto& t : b)
                    if (t.state == T_FLIGHT && alive == 0) return FSQ_EINTERNAL;
            }
        }
        if (alive_out) *alive_out = alive;
        if (finished_out) *finished_out = fin_total;
        return FSQ_OK;
    }

    ~FsqFitQueue()
    {
        if (hi) hi_release(hi);
        for (auto& t : b)
            if (t.ev) (void)hipEventDestroy(t.ev);
        if (owned_ws) (void)hipFree(owned_ws);
    }
};

extern "C" int64_t fsq_fit_workspace_bytes(int64_t n)
{
    if (n < 0) return FSQ_EINVAL;
    return (int64_t)layout_bytes((size_t)n + 64, (size_t)n + 64);
}

// One stand-alone batch: a temporary engine in the caller's workspace.  Returns when the last round has run; the
// row-writing kernel is enqueued on the caller's stream.
int fsq_launch_fit_rounds(const uint16_t* d_src, int H, int W, const int32_t* d_cand, int64_t n, int mode, bool from_image,
                          FsqRow* d_rows, void* d_ws, int64_t ws_bytes, hipStream_t s_user)
{
    if (n > 2000000000ll) return FSQ_ENOTIMPL;
    if (ws_bytes < fsq_fit_workspace_bytes(n) || !d_ws) return FSQ_ENOMEM;
    FsqFitQueue q;
    int rc = q.init(d_ws, ws_bytes, (size_t)n + 64, (size_t)n + 64, mode, s_user, true);
    if (rc != FSQ_OK) return rc;
    rc = q.submit(d_src, (mode & FSQ_PIXELS_U32_FLAG) ? FSQ_PIXELS_U32 : (mode & FSQ_PIXELS_F16_FLAG) ? FSQ_PIXELS_F16 : FSQ_PIXELS_U16, H, W, d_cand, n, from_image,
                  d_rows, nullptr);
    if (rc != FSQ_OK) return rc;
    while (q.alive > 0) {
        rc = q.advance(q.cfg.max_rounds, 0, nullptr, nullptr);
        if (rc != FSQ_OK) return rc;
        if (q.cfg.max_rounds > 0) {             // debug: stop early, rows of unfinished fits are undefined
            FSQ_HIP_CHECK(hipStreamSynchronize(q.s));
            break;
        }
    }
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}

// ---- FsqFitQueue: the engine kept alive across batches (include/fsq.h) ------------------------------------------
extern "C" int64_t fsq_fitq_workspace_bytes(int64_t pool_slots, int64_t queue_cap)
{
    if (pool_slots <= 0 || queue_cap <= 0) return FSQ_EINVAL;
    return (int64_t)layout_bytes((size_t)pool_slots, (size_t)queue_cap);
}

extern "C" int fsq_fitq_create(FsqFitQueue** out, void* d_workspace, int64_t workspace_bytes, int64_t pool_slots,
                               int64_t queue_cap, int mode, void* stream)
{
    if (!out || pool_slots <= 0 || queue_cap <= 0) return FSQ_EINVAL;
    // mode | FSQ_PIXELS_U32_FLAG: a queue for batches of uint32 pixels (and for nothing else: the compact ROI copies of all
    // batches in flight share one word size); the two 16-bit formats may be mixed in one queue
    const int m = mode & ~FSQ_PIXELS_U32_FLAG;
    if (m != FSQ_MODE_REF && m != FSQ_MODE_TEXTBOOK && m != FSQ_MODE_TEXTBOOK_F32) return FSQ_EINVAL;
    if ((mode & FSQ_PIXELS_U32_FLAG) && m == FSQ_MODE_TEXTBOOK_F32) return FSQ_ENOTIMPL;
    FsqFitQueue* q = new (std::nothrow) FsqFitQueue();
    if (!q) return FSQ_ENOMEM;
    int rc = q->init(d_workspace, workspace_bytes, (size_t)pool_slots, (size_t)queue_cap, mode, (hipStream_t)stream, false);
    if (rc != FSQ_OK) { delete q; return rc; }
    *out = q;
    return FSQ_OK;
}

extern "C" int fsq_fitq_submit(FsqFitQueue* q, const void* d_img, int pixel_format, int n_fields, int H, int W,
                               const int32_t* d_cand, int64_t n, FsqRow* d_rows, int* ticket)
{
    if (!q || n_fields < 0 || H < 5 || W < 5) return FSQ_EINVAL;
    return q->submit((const uint16_t*)d_img, pixel_format, H, W, d_cand, n, true, d_rows, ticket);
}

extern "C" int fsq_fitq_advance(FsqFitQueue* q, int64_t max_rounds, int64_t alive_below, int64_t* alive, int* finished)
{
    if (!q) return FSQ_EINVAL;
    long long a = 0;
    int rc = q->advance(max_rounds, alive_below, &a, finished);
    if (alive) *alive = a;
    return rc;
}

extern "C" int fsq_fitq_take(FsqFitQueue* q, int ticket, void* consumer_stream)
{
    if (!q || ticket < 0 || ticket >= FSQ_MAX_TICKETS || q->b[ticket].state == T_FREE) return FSQ_EINVAL;
    Batch& B = q->b[ticket];
    if (B.state != T_FINISHED) return 0;
    if (B.ev && (hipStream_t)consumer_stream != q->s_finish) FSQ_HIP_CHECK(hipStreamWaitEvent((hipStream_t)consumer_stream, B.ev, 0));
    B.state = T_FREE;
    return 1;
}

extern "C" int64_t fsq_fitq_alive(const FsqFitQueue* q) { return q ? q->alive : FSQ_EINVAL; }
extern "C" int64_t fsq_fitq_rounds(const FsqFitQueue* q) { return q ? q->round : FSQ_EINVAL; }

extern "C" int fsq_fitq_destroy(FsqFitQueue* q)
{
    if (!q) return FSQ_EINVAL;
    hipError_t e = hipStreamSynchronize(q->s);
    delete q;
    if (e != hipSuccess) { g_fsq_last_hip = e; return FSQ_EHIP; }
    return FSQ_OK;
}
