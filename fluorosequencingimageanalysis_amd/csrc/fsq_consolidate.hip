// fsq_consolidate.hip - K5: R^2 filter, consolidation of competing PSFs and re-keying, per field.
// Reference: pflib.find_peptides, pflib.py:466 (filter), 477-512 (consolidation), 514-519 (re-key).
//
// The reference's loops are sequential and order dependent (raster order of the candidate pixels, raster order inside each
// search window): a candidate that still has its dict entry scans the (2r+5)^2 window around it, deleting every rival
// with a smaller R^2 until the first rival whose R^2 is not smaller - then the candidate itself is deleted and the scan
// stops.  What one candidate's turn reads and writes lies inside its own window, so two candidates whose windows do not
// overlap - more than 2 (r + 2) pixels apart in h or in w - commute, and the sequential result is reproduced exactly by
// ANY schedule in which a candidate takes its turn after every raster-earlier candidate whose window overlaps its own.
// Round 4: one BLOCK of 8 or 16 waves owns a field.  Wave v takes the surviving candidates v, v + NW, ... in raster order;
// before a candidate's turn the wave looks at the raster-earlier half of the (4r+9)^2 neighbourhood and waits until no
// entry alive there that has not had its turn (a flag per candidate) can still touch this candidate's window - it lies in
// the window itself or shares a living entry with it; the turn itself - the window scan, done by the lanes in parallel
// and resolved with ballots so that it equals the reference's sequential scan - is the round-1 code.  The candidate with
// the smallest number that has not had its turn never waits, so the scheme cannot deadlock.  (Round 3 walked a field with
// ONE wave: 21-30 ms for a 2 048^2 field of 46 000 candidates whatever the number of fields.)
// A pixel -> candidate grid (int32 per pixel, workspace) plays the role of the reference's dict: -1 empty, i >= 0 the
// entry of candidate i, -(i + 2) the deleted entry of candidate i.
#include <algorithm>
#include <cstdlib>

#include "fsq_common.h"
#include "fsq_devmath.h"

namespace {

__device__ __forceinline__ double round_key(double x, int py2) { return py2 ? round(x) : rint(x); }

// loads / stores other waves of the block act upon while this one runs: atomic at WORKGROUP scope - all waves of a block sit on
// one CU and share its vector L1, so nothing has to leave the CU (agent-scope fences write back and invalidate the whole L2
// of the XCD on gfx950: tried first, 50 ms per launch)
__device__ __forceinline__ int ld_shared_i(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_shared_i(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int ld_shared_b(const unsigned char* p) { return (int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(v, d); v += (lane >= d) ? t : 0; }
    return v;
}

// Ordered compaction over the candidates 0 .. cnt-1 of a field by all waves of the block: out[out_base + rank(i)] = value(i) for
// every i with pred(i), ranks in index order.  chunkc: cnt / 64 + 1 ints of scratch.  Returns the number selected.
template <typename Pred, typename Val>
__device__ int block_compact(int cnt, int* chunkc, int* out, int out_base, Pred pred, Val value, int* s_total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    const int nchunk = (cnt + 63) >> 6;
    for (int c = wave; c < nchunk; c += NW) {
        const int i = c * 64 + lane;
        const unsigned long long m = __ballot(i < cnt && pred(i));
        if (lane == 0) chunkc[c] = __popcll(m);
    }
    __syncthreads();
    if (wave == 0) {
        int run = 0;
        for (int base = 0; base < nchunk; base += 64) {
            const int c = base + lane;
            const int v = c < nchunk ? chunkc[c] : 0;
            const int incl = wave_incl_scan(v, lane);
            if (c < nchunk) chunkc[c] = run + incl - v;
            run += __shfl(incl, 63);
        }
        if (lane == 0) *s_total = run;
    }
    __syncthreads();
    for (int c = wave; c < nchunk; c += NW) {
        const int i = c * 64 + lane;
        const bool ok = i < cnt && pred(i);
        const unsigned long long m = __ballot(ok);
        if (ok) out[out_base + chunkc[c] + __popcll(m & ((1ull << lane) - 1ull))] = value(i);
    }
    __syncthreads();
    return *s_total;
}

// grid = n_fields blocks of NW waves (one block per field)
__global__ void __launch_bounds__(1024) k5_consolidate(FsqRow* __restrict__ rows, const int* __restrict__ counts,
                                                       const int* __restrict__ offsets, int H, int W, double r2_thr,
                                                       int radius, int py2, int* __restrict__ grid_all,
                                                       unsigned char* __restrict__ turn_all, int* __restrict__ chunk_all, long long chunk_stride,
                                                       int* __restrict__ keep, int* __restrict__ nkeep)
{
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    const int off = offsets[f], cnt = counts[f];
    int* grid = grid_all + (size_t)f * H * W;          // pre-set to -1
    unsigned char* turn = turn_all + (size_t)f * H * W; // by candidate: 1 once the candidate has had its turn
    int* chunkc = chunk_all + (size_t)f * chunk_stride;
    FsqRow* R = rows + off;
    const double rr = (double)(radius * radius);
    __shared__ int s_total, s_assert;

    // dict insertion (setdefault, pflib.py:477): survivors of the R^2 filter (NaN passes, :466).  Their indices are also
    // compacted, in order, into this field's slice of `keep` (scratch until the final list is written): the walks below
    // then touch ~15 % of the candidates.
    int* surv = keep + off;
    for (int i = tid; i < cnt; i += blockDim.x) {
        R[i].key_h = -1; R[i].key_w = -1;
        if (!(R[i].r2 < r2_thr)) { grid[(size_t)R[i].h * W + R[i].w] = i; turn[i] = 0; }
    }
    __syncthreads();
    const int nsurv = block_compact(cnt, chunkc, keep, off, [&](int i) { return !(R[i].r2 < r2_thr); }, [](int i) { return i; }, &s_total);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();

    // consolidation, pflib.py:479-512
    const int D = 2 * (radius + 2);                     // windows overlap up to this distance
    // Which wave takes which survivor: wave v owns the v-th vertical STRIP of the field and walks the survivors of its strip
    // in raster order.  (Dealing the survivors out in turn - wave v takes survivor v, v + NW, ... - makes neighbours in
    // raster order, which are neighbours in the image, wait for each other all the time: measured, 3 of 8 waves busy.)
    // With strips, only the candidates within 2 (r + 2) pixels of a strip border depend on another wave at all, and that
    // wave has usually passed the row in question.  Every wave reads the whole survivor list, 64 entries at a time.
    const int strip_w = (W + NW - 1) / NW;
    for (int cbase = 0; cbase < nsurv; cbase += 64) {
      int c_i = -1, c_h = 0, c_w = 0;
      double c_h0 = 0., c_w0 = 0., c_r2 = 0.;
      if (cbase + lane < nsurv) {
          c_i = surv[cbase + lane];
          c_h = R[c_i].h; c_w = R[c_i].w; c_h0 = R[c_i].h0; c_w0 = R[c_i].w0; c_r2 = R[c_i].r2;
      }
#ifdef FSQ_EXPERIMENT_NO_TURNS              // timing experiment (results wrong): the kernel without the consolidation turns
      unsigned long long mine = 0ull;
#else
      unsigned long long mine = __ballot(c_i >= 0 && (c_w / strip_w) == wave);
#endif
      while (mine) {
        const int src = __ffsll((long long)mine) - 1;
        mine &= mine - 1;
        const int i = __shfl(c_i, src), h = __shfl(c_h, src), w = __shfl(c_w, src);
        const double h0 = __shfl(c_h0, src), w0 = __shfl(c_w0, src), r2i = __shfl(c_r2, src);
        // The turn: (a) wait for the raster-earlier entries that can still change what this turn sees, (b) the window scan.
        // An earlier entry j that has not had its turn can only touch this candidate's window if it lies in it itself, or
        // if some living entry lies in BOTH windows (j's turn deletes j itself or rivals inside j's window; entries never
        // appear during consolidation) - the same holds the other way round for the later entries, so whatever this turn
        // reads is final once no such j is left.  Everything is loaded NG x 64 cells at a time with all loads of a group in
        // flight together: a turn is a handful of dependent memory round trips and nothing else.
        constexpr int NG = 5, NGW = 3;      // (defaults: 312 raster-earlier cells of the 25 x 25 neighbourhood, 13 x 13 window)
        const int rw = radius + 2;
        const int h_lo = max(0, h - rw), h_hi = min(h + rw + 1, H), w_lo = max(0, w - rw), w_hi = min(w + rw + 1, W);
        const int ww = w_hi - w_lo, ncell = (h_hi - h_lo) * ww;
        const bool small_window = ncell <= 64 * NGW;               // (radius <= 4: the window fits the registers of one pass)
        const int nh_lo = max(0, h - D), nw_lo = max(0, w - D), nw_hi = min(W, w + D + 1);
        const int nww = nw_hi - nw_lo, nearly = (h - nh_lo) * nww + (w - nw_lo);
        int wk[NGW];                        // the window's cells (small windows), as loaded by the last look
        bool wk_valid = false;
        while (true) {
            // (1) the raster-earlier entries of the neighbourhood that are alive and have not had their turn: their pixels
            // are noted (lane q keeps the q-th one); (2) only THEN the window is read - an entry that finished its turn before
            // (1) has left its mark in what (2) sees, one that is noted in (1) is judged against what (2) sees: whatever it
            // still shares with this window is alive there
            int pend_h = 0, pend_w = 0, npend = 0;
            for (int base = 0; base < nearly; base += 64 * NG) {
                int kk[NG], tt[NG];
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const int c = base + 64 * g + lane;
                    kk[g] = (c < nearly) ? ld_shared_i(&grid[(size_t)(nh_lo + c / nww) * W + (nw_lo + c % nww)]) : -1;
                }
#pragma unroll
                for (int g = 0; g < NG; g++) tt[g] = (kk[g] >= 0) ? ld_shared_b(&turn[kk[g]]) : 1;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    unsigned long long m = __ballot(tt[g] == 0);
                    while (m) {
                        const int c = base + 64 * g + (__ffsll((long long)m) - 1);
                        m &= m - 1;
                        if (lane == (npend & 63)) { pend_h = nh_lo + c / nww; pend_w = nw_lo + c % nww; }
                        npend++;
                    }
                }
            }
            wk_valid = false;
#ifdef FSQ_EXPERIMENT_NO_WAIT               // timing experiment (results wrong): nobody waits for anybody
            npend = 0;
#endif
            if (npend == 0) break;
            bool blocked = !small_window || npend > 64;
            if (!blocked) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
                for (int g = 0; g < NGW; g++) {
                    const int c = 64 * g + lane;
                    wk[g] = -1;
                    if (c < ncell) {
                        const int hd = h_lo + c / ww, wd = w_lo + c % ww;
                        if (!(hd == h && wd == w)) wk[g] = ld_shared_i(&grid[(size_t)hd * W + wd]);
                    }
                }
                wk_valid = true;
                for (int q = 0; q < npend && !blocked; q++) {
                    const int hj = __shfl(pend_h, q), wj = __shfl(pend_w, q);
                    if (abs(hj - h) <= rw && abs(wj - w) <= rw) blocked = true;        // it lies in this window itself
                    else {
                        bool both = false;
#pragma unroll
                        for (int g2 = 0; g2 < NGW; g2++) {
                            const int c2 = 64 * g2 + lane;
                            both = both || (wk[g2] >= 0 && abs(h_lo + c2 / ww - hj) <= rw && abs(w_lo + c2 % ww - wj) <= rw);
                        }
                        if (__ballot(both)) blocked = true;                             // a living entry lies in both windows
                    }
                }
            }
            if (!blocked) break;
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        {
            const int own = ld_shared_i(&grid[(size_t)h * W + w]);     // (issued together with the window's first cells)
            bool dead = false;
            for (int base = 0; base < ncell && !dead; base += 64 * NGW) {
                int kk[NGW];
                double rh0[NGW], rw0[NGW], rr2[NGW];
#pragma unroll
                for (int g = 0; g < NGW; g++) {
                    const int c = base + 64 * g + lane;
                    kk[g] = -1;
                    if (wk_valid) kk[g] = wk[g];
                    else if (c < ncell) {
                        const int hd = h_lo + c / ww, wd = w_lo + c % ww;
                        if (!(hd == h && wd == w)) kk[g] = ld_shared_i(&grid[(size_t)hd * W + wd]);
                    }
                }
                if (own != i) break;                                    // deleted by an earlier candidate: nothing to do
#pragma unroll
                for (int g = 0; g < NGW; g++) {
                    const int k = kk[g] >= 0 ? kk[g] : i;
                    rh0[g] = R[k].h0; rw0[g] = R[k].w0; rr2[g] = R[k].r2;
                }
#pragma unroll
                for (int g = 0; g < NGW; g++) {
                    if (dead || base + 64 * g >= ncell) continue;       // (wave-uniform)
                    const int c = base + 64 * g + lane;
                    bool rival = false, lose = false;
                    if (kk[g] >= 0) {
                        const double dh = h0 - rh0[g], dw = w0 - rw0[g];
                        if (!(fsq_pow2(dh) + fsq_pow2(dw) > rr)) {       // numpy scalar **2, pflib.py:505
                            rival = true;
                            lose = !(r2i > rr2[g]);                     // pflib.py:508
                        }
                    }
                    const unsigned long long mlose = __ballot(lose);
                    const int first = mlose ? (__ffsll((long long)mlose) - 1) : 64;
                    if (rival && lane < first) st_shared_i(&grid[(size_t)(h_lo + c / ww) * W + (w_lo + c % ww)], -(kk[g] + 2));     // rivals with smaller R^2 die
                    if (mlose) {
                        if (lane == 0) st_shared_i(&grid[(size_t)h * W + w], -(i + 2));     // the candidate itself dies, scan stops
                        dead = true;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                                // the turn's deletions are visible before the flag is
        if (lane == 0) __hip_atomic_store(&turn[i], (unsigned char)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (tid == 0) s_assert = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();

    // re-key, pflib.py:514-519: every entry still alive whose rounded centre differs from its pixel moves to the rounded
    // key, in index order, and the reference asserts that the new key is free AT THAT MOMENT.  No assertion fires iff
    // (1) a moving entry's target holds, at the start, nothing or an entry that itself moves away EARLIER (smaller index) and
    // (2) no two moving entries have the same target; if one fires the field has no table at all (nkeep = -1), so only
    // whether one fires is reproduced - by all threads at once: check (1) on the untouched grid, vacate, land with an
    // exchange that reports a second arrival.
    auto key_of = [&](int i, int* hr, int* wr) { *hr = (int)round_key(R[i].h0, py2); *wr = (int)round_key(R[i].w0, py2); };
    for (int s_ = tid; s_ < nsurv; s_ += blockDim.x) {
        const int i = surv[s_];
        if (grid[(size_t)R[i].h * W + R[i].w] != i) continue;
        int hr, wr;
        key_of(i, &hr, &wr);
        R[i].key_h = hr; R[i].key_w = wr;
        if ((hr != R[i].h || wr != R[i].w) && hr >= 0 && hr < H && wr >= 0 && wr < W) {
            const int j = grid[(size_t)hr * W + wr];
            if (j >= 0) {
                int hj, wj;
                key_of(j, &hj, &wj);
                if (!(j < i && (hj != R[j].h || wj != R[j].w))) s_assert = 1;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    for (int s_ = tid; s_ < nsurv; s_ += blockDim.x) {
        const int i = surv[s_];
        if (R[i].key_h >= 0 && (R[i].key_h != R[i].h || R[i].key_w != R[i].w)) grid[(size_t)R[i].h * W + R[i].w] = -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    for (int s_ = tid; s_ < nsurv; s_ += blockDim.x) {
        const int i = surv[s_], hr = R[i].key_h, wr = R[i].key_w;
        if (hr >= 0 && (hr != R[i].h || wr != R[i].w) && hr < H && wr >= 0 && wr < W)
            if (atomicExch(&grid[(size_t)hr * W + wr], i) >= 0) s_assert = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();

    // kept list in the reference's dict order: untouched keys in insertion order, re-keyed ones appended
    const int nk0 = block_compact(cnt, chunkc, keep, off, [&](int i) { return R[i].key_h >= 0 && R[i].key_h == R[i].h && R[i].key_w == R[i].w; },
                                  [&](int i) { return off + i; }, &s_total);
    const int nk1 = block_compact(cnt, chunkc, keep, off + nk0, [&](int i) { return R[i].key_h >= 0 && (R[i].key_h != R[i].h || R[i].key_w != R[i].w); },
                                  [&](int i) { return off + i; }, &s_total);
    if (tid == 0) nkeep[f] = s_assert ? -1 : nk0 + nk1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4, second form (the default): CONNECTED COMPONENTS.  Two candidates can only influence each other's turns if their
// windows overlap - |dh| <= 2 (r + 2) and |dw| <= 2 (r + 2) - so the survivors fall into components of that relation which never
// touch each other's cells, and within a component the turns are taken in raster order by ONE wave: no wave ever waits for
// another, and the components of all fields (a 2 048^2 field has thousands: one survivor per ~600 pixels) spread over the
// whole chip instead of over the 8 or 16 waves of one block.
//   k5c_insert   filter (pflib.py:466), dict insertion into the pixel -> candidate grid, parent[i] = i
//   k5c_union    every survivor looks at the raster-earlier half of its (4r+9)^2 neighbourhood and unites itself with every
//                survivor it finds (lock-free union-find: roots are linked under smaller indices with a compare-and-swap,
//                finds halve paths; all accesses to `parent` are relaxed device-scope atomics - no fence anywhere)
//   k5c_flatten  parent[i] = root of i (the smallest index of the component), last[root] = its largest index
//   k5c_turns    the wave that meets a root walks root .. last[root], picks the members out by their label and gives each
//                its turn (the window scan of the block kernel above, without the waiting)
//   k5c_finish   re-key + kept list per field (the block kernel's tail)
__device__ __forceinline__ int ldA(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stA(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ int uf_find(int* P, int x)
{
    for (;;) {
        const int p = ldA(P + x);
        if (p == x) return x;
        const int gp = ldA(P + p);
        if (gp == p) return p;
        stA(P + x, gp);                 // path halving (a racing link of x cannot be lost: x is not a root here)
        x = gp;
    }
}
// the same walk with ordinary loads: what it sees may be out of date (another XCD's links are not in this one's L2) - an older state
// of a forest whose pointers only ever move towards smaller indices, so the node it ends at is a member of the same tree, which is all a
// link needs; only the link itself has to be atomic
__device__ int uf_find_plain(const int* P, int x)
{
    for (;;) {
        const int p = P[x];
        if (p == x) return x;
        x = p;
    }
}
__device__ void uf_union(int* P, int a, int b)
{
    a = uf_find_plain(P, a); b = uf_find_plain(P, b);    // first try on cached values (a device-scope atomic load is a trip to memory)
    for (;;) {
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        if (atomicCAS(P + a, a, b) == a) return;        // the larger root goes under the smaller one: parent[x] <= x always, no cycles
        a = uf_find(P, a); b = uf_find(P, b);           // a was no root any more: look again, this time at memory
    }
}

struct K5Field { int off, cnt; FsqRow* R; int* grid; int* parent; int* last; };
__device__ __forceinline__ K5Field k5_field(FsqRow* rows, const int* counts, const int* offsets, int f, int H, int W, int* ws, long long stride)
{
    K5Field F;
    F.off = offsets[f]; F.cnt = counts[f]; F.R = rows + F.off;
    int* base = ws + (size_t)f * stride;
    F.grid = base; F.parent = base + (size_t)H * W; F.last = base + 2 * (size_t)H * W;
    return F;
}

// grid = (blocks per field, n_fields), block = 64
__global__ void __launch_bounds__(64) k5c_insert(FsqRow* __restrict__ rows, const int* __restrict__ counts, const int* __restrict__ offsets,
                                                 int H, int W, double r2_thr, int* __restrict__ ws, long long stride)
{
    const K5Field F = k5_field(rows, counts, offsets, blockIdx.y, H, W, ws, stride);
    for (int i = blockIdx.x * 64 + threadIdx.x; i < F.cnt; i += gridDim.x * 64) {
        F.R[i].key_h = -1; F.R[i].key_w = -1;
        const bool surv = !(F.R[i].r2 < r2_thr);
        F.parent[i] = surv ? i : -1;
        F.last[i] = i;
        if (surv) F.grid[(size_t)F.R[i].h * W + F.R[i].w] = i;       // pre-set to -1
    }
}

__global__ void __launch_bounds__(64) k5c_union(FsqRow* __restrict__ rows, const int* __restrict__ counts, const int* __restrict__ offsets,
                                                int H, int W, int radius, int* __restrict__ ws, long long stride)
{
    const K5Field F = k5_field(rows, counts, offsets, blockIdx.y, H, W, ws, stride);
    const int lane = threadIdx.x, D = 2 * (radius + 2);
    for (int cbase = blockIdx.x * 64; cbase < F.cnt; cbase += gridDim.x * 64) {
        const int ci = cbase + lane;
        int c_h = 0, c_w = 0;
        bool surv = false;
        if (ci < F.cnt) { surv = ldA(F.parent + ci) >= 0; c_h = F.R[ci].h; c_w = F.R[ci].w; }
        unsigned long long m = __ballot(surv);
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int i = cbase + src, h = __shfl(c_h, src), w = __shfl(c_w, src);
            const int nh_lo = max(0, h - D), nw_lo = max(0, w - D), nw_hi = min(W, w + D + 1);
            const int nww = nw_hi - nw_lo, nearly = (h - nh_lo) * nww + (w - nw_lo);       // the raster-earlier cells of the neighbourhood
            for (int base = 0; base < nearly; base += 64 * 5) {
                int kk[5];
#pragma unroll
                for (int g = 0; g < 5; g++) {
                    const int c = base + 64 * g + lane;
                    kk[g] = (c < nearly) ? F.grid[(size_t)(nh_lo + c / nww) * W + (nw_lo + c % nww)] : -1;
                }
#pragma unroll
                for (int g = 0; g < 5; g++)
                    if (kk[g] >= 0) uf_union(F.parent, i, kk[g]);
            }
        }
    }
}

__global__ void __launch_bounds__(64) k5c_flatten(const int* __restrict__ counts, const int* __restrict__ offsets, int H, int W,
                                                  int* __restrict__ ws, long long stride)
{
    const int f = blockIdx.y, cnt = counts[f];
    int* base = ws + (size_t)f * stride;
    int* parent = base + (size_t)H * W;
    int* last = base + 2 * (size_t)H * W;
    for (int i = blockIdx.x * 64 + threadIdx.x; i < cnt; i += gridDim.x * 64) {
        if (ldA(parent + i) < 0) continue;
        int r = i;
        for (;;) { const int p = ldA(parent + r); if (p == r) break; r = p; }      // (no more links are made: the chains only get shorter)
        stA(parent + i, r);
        atomicMax(last + r, i);
    }
}

__global__ void __launch_bounds__(64) k5c_turns(FsqRow* __restrict__ rows, const int* __restrict__ counts, const int* __restrict__ offsets,
                                                int H, int W, int radius, int* __restrict__ ws, long long stride)
{
    const K5Field F = k5_field(rows, counts, offsets, blockIdx.y, H, W, ws, stride);
    const int lane = threadIdx.x, rw = radius + 2;
    const double rr = (double)(radius * radius);
    const FsqRow* R = F.R;
    int* grid = F.grid;
    for (int cbase = blockIdx.x * 64; cbase < F.cnt; cbase += gridDim.x * 64) {
        const int ci = cbase + lane;
        unsigned long long roots = __ballot(ci < F.cnt && F.parent[ci] == ci);
        while (roots) {
            const int root = cbase + __ffsll((long long)roots) - 1;
            roots &= roots - 1;
            const int last = F.last[root];
            if (last == root) continue;                         // a component of one: its turn finds no rival (the common case)
            for (int mb = root & ~63; mb <= last; mb += 64) {   // the members, in raster order
                const int mj = mb + lane;
                int m_h = 0, m_w = 0;
                double m_h0 = 0., m_w0 = 0., m_r2 = 0.;
                bool mem = false;
                if (mj >= root && mj <= last && F.parent[mj] == root) {
                    mem = true;
                    m_h = R[mj].h; m_w = R[mj].w; m_h0 = R[mj].h0; m_w0 = R[mj].w0; m_r2 = R[mj].r2;
                }
                unsigned long long mm = __ballot(mem);
                while (mm) {
                    const int src = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const int i = mb + src, h = __shfl(m_h, src), w = __shfl(m_w, src);
                    const double h0 = __shfl(m_h0, src), w0 = __shfl(m_w0, src), r2i = __shfl(m_r2, src);
                    if (ld_shared_i(&grid[(size_t)h * W + w]) != i) continue;      // deleted by an earlier member
                    const int h_lo = max(0, h - rw), h_hi = min(h + rw + 1, H), w_lo = max(0, w - rw), w_hi = min(w + rw + 1, W);
                    const int ww = w_hi - w_lo, ncell = (h_hi - h_lo) * ww;
                    bool dead = false;
                    constexpr int NGW = 3;
                    for (int base = 0; base < ncell && !dead; base += 64 * NGW) {
                        int kk[NGW];
                        double rh0[NGW], rw0[NGW], rr2[NGW];
#pragma unroll
                        for (int g = 0; g < NGW; g++) {
                            const int c = base + 64 * g + lane;
                            kk[g] = -1;
                            if (c < ncell) {
                                const int hd = h_lo + c / ww, wd = w_lo + c % ww;
                                if (!(hd == h && wd == w)) kk[g] = ld_shared_i(&grid[(size_t)hd * W + wd]);
                            }
                        }
#pragma unroll
                        for (int g = 0; g < NGW; g++) {
                            const int k = kk[g] >= 0 ? kk[g] : i;
                            rh0[g] = R[k].h0; rw0[g] = R[k].w0; rr2[g] = R[k].r2;
                        }
#pragma unroll
                        for (int g = 0; g < NGW; g++) {
                            if (dead || base + 64 * g >= ncell) continue;       // (wave-uniform)
                            const int c = base + 64 * g + lane;
                            bool rival = false, lose = false;
                            if (kk[g] >= 0) {
                                const double dh = h0 - rh0[g], dw = w0 - rw0[g];
                                if (!(fsq_pow2(dh) + fsq_pow2(dw) > rr)) {       // numpy scalar **2, pflib.py:505
                                    rival = true;
                                    lose = !(r2i > rr2[g]);                     // pflib.py:508
                                }
                            }
                            const unsigned long long mlose = __ballot(lose);
                            const int first = mlose ? (__ffsll((long long)mlose) - 1) : 64;
                            if (rival && lane < first) st_shared_i(&grid[(size_t)(h_lo + c / ww) * W + (w_lo + c % ww)], -(kk[g] + 2));
                            if (mlose) {
                                if (lane == 0) st_shared_i(&grid[(size_t)h * W + w], -(i + 2));
                                dead = true;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");     // this turn's deletions before the next member's reads
                }
            }
        }
    }
}

// re-key + kept list (pflib.py:514-519): one block per field - the tail of the block kernel above, over all candidates
__global__ void __launch_bounds__(512) k5c_finish(FsqRow* __restrict__ rows, const int* __restrict__ counts, const int* __restrict__ offsets,
                                                  int H, int W, int py2, int* __restrict__ ws, long long stride, int* __restrict__ chunk_all,
                                                  long long chunk_stride, int* __restrict__ keep, int* __restrict__ nkeep)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    const K5Field F = k5_field(rows, counts, offsets, f, H, W, ws, stride);
    FsqRow* R = F.R;
    int* grid = F.grid;
    const int cnt = F.cnt, off = F.off;
    int* chunkc = chunk_all + (size_t)f * chunk_stride;
    __shared__ int s_total, s_assert;
    if (tid == 0) s_assert = 0;
    __syncthreads();
    auto key_of = [&](int i, int* hr, int* wr) { *hr = (int)round_key(R[i].h0, py2); *wr = (int)round_key(R[i].w0, py2); };
    for (int i = tid; i < cnt; i += blockDim.x) {
        if (F.parent[i] < 0 || grid[(size_t)R[i].h * W + R[i].w] != i) continue;     // not a survivor / deleted
        int hr, wr;
        key_of(i, &hr, &wr);
        R[i].key_h = hr; R[i].key_w = wr;
        if ((hr != R[i].h || wr != R[i].w) && hr >= 0 && hr < H && wr >= 0 && wr < W) {
            const int j = grid[(size_t)hr * W + wr];
            if (j >= 0) {
                int hj, wj;
                key_of(j, &hj, &wj);
                if (!(j < i && (hj != R[j].h || wj != R[j].w))) s_assert = 1;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    for (int i = tid; i < cnt; i += blockDim.x)
        if (R[i].key_h >= 0 && (R[i].key_h != R[i].h || R[i].key_w != R[i].w)) grid[(size_t)R[i].h * W + R[i].w] = -1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    for (int i = tid; i < cnt; i += blockDim.x) {
        const int hr = R[i].key_h, wr = R[i].key_w;
        if (hr >= 0 && (hr != R[i].h || wr != R[i].w) && hr < H && wr >= 0 && wr < W)
            if (atomicExch(&grid[(size_t)hr * W + wr], i) >= 0) s_assert = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    const int nk0 = block_compact(cnt, chunkc, keep, off, [&](int i) { return R[i].key_h >= 0 && R[i].key_h == R[i].h && R[i].key_w == R[i].w; },
                                  [&](int i) { return off + i; }, &s_total);
    const int nk1 = block_compact(cnt, chunkc, keep, off + nk0, [&](int i) { return R[i].key_h >= 0 && (R[i].key_h != R[i].h || R[i].key_w != R[i].w); },
                                  [&](int i) { return off + i; }, &s_total);
    if (tid == 0) nkeep[f] = s_assert ? -1 : nk0 + nk1;
}

__global__ void k5_total(int* __restrict__ nkeep, int n_fields)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int t = 0;
        for (int f = 0; f < n_fields; f++) t += nkeep[f] > 0 ? nkeep[f] : 0;
        nkeep[n_fields] = t;
    }
}

// Kept rows of all fields, contiguous and in field order (the peak table that is gathered / copied to the host).
// One block per field; the field's position in the output is the sum of the kept counts before it.
__global__ void __launch_bounds__(256) k5_kept_rows(const FsqRow* __restrict__ rows, const int* __restrict__ keep,
                                                    const int* __restrict__ offsets, const int* __restrict__ nkeep,
                                                    FsqRow* __restrict__ out, int* __restrict__ out_offsets, long long cap)
{
    __shared__ int s_part[256];
    const int f = blockIdx.x, t = threadIdx.x;
    int acc = 0;
    for (int g = t; g < f; g += 256) acc += nkeep[g] > 0 ? nkeep[g] : 0;
    s_part[t] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) s_part[t] += s_part[t + w];
        __syncthreads();
    }
    const int base = s_part[0], nk = nkeep[f] > 0 ? nkeep[f] : 0, off = offsets[f];
    if (t == 0) out_offsets[f] = base;
    if (t == 0 && f == (int)gridDim.x - 1) out_offsets[f + 1] = base + nk;
    // a row is 128 bytes = 8 x 16 bytes: 8 consecutive threads move one row
    const uint4* src = (const uint4*)rows;
    uint4* dst = (uint4*)out;
    for (int k = t >> 3; k < nk; k += 32) {
        if ((long long)base + k >= cap) break;
        const int r = keep[off + k];
        dst[((size_t)base + k) * 8 + (t & 7)] = src[(size_t)r * 8 + (t & 7)];
    }
}

}  // namespace

extern "C" int fsq_kept_rows(const FsqRow* d_rows, const int32_t* d_keep, const int32_t* d_offsets, const int32_t* d_nkeep,
                             int n_fields, FsqRow* d_out, int64_t cap, int32_t* d_out_offsets, void* stream)
{
    if (n_fields < 1 || !d_rows || !d_keep || !d_offsets || !d_nkeep || (!d_out && cap > 0) || !d_out_offsets || cap < 0) return FSQ_EINVAL;
    hipLaunchKernelGGL(k5_kept_rows, dim3(n_fields), dim3(256), 0, (hipStream_t)stream, d_rows, d_keep, d_offsets, d_nkeep,
                       d_out, d_out_offsets, (long long)cap);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}

// per field: the pixel -> candidate grid, the union-find parents and the components' last members (int32 per pixel each: there is at
// most one candidate per pixel), and the chunk counters of the ordered compactions (one per 64 candidates)
static int64_t k5_chunk_stride(int H, int W) { return ((int64_t)H * W + 63) / 64 + 2; }
extern "C" int64_t fsq_consolidate_workspace_bytes(int n_fields, int H, int W)
{
    if (n_fields < 1 || H < 1 || W < 1) return FSQ_EINVAL;
    return (int64_t)n_fields * ((int64_t)H * W * 12 + k5_chunk_stride(H, W) * 4) + 256;
}

extern "C" int fsq_consolidate(FsqRow* d_rows, const int32_t* d_counts, const int32_t* d_offsets, int n_fields, int H,
                               int W, double r2_threshold, int radius, int py2_round, int32_t* d_keep, int32_t* d_nkeep,
                               void* d_workspace, int64_t workspace_bytes, void* stream)
{
    if (radius < 2) return FSQ_EINVAL;                                // pflib.py:431-432 -> ValueError
    if (n_fields < 1 || H < 5 || W < 5 || !d_rows || !d_counts || !d_offsets || !d_keep || !d_nkeep || !d_workspace) return FSQ_EINVAL;
    if (workspace_bytes < fsq_consolidate_workspace_bytes(n_fields, H, W)) return FSQ_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const size_t px = (size_t)H * W;
    int* ws = (int*)d_workspace;
    const long long stride = 3 * (long long)px;                       // ints per field: grid | parent | last
    int* chunks = ws + (size_t)n_fields * stride;
    // Which form: both give the same tables.  The components form spreads a field over the whole chip; once the FIELDS alone fill
    // it (1 024 fields of 512^2: 8 waves each) the block form, which scans a neighbourhood only when a candidate has to wait, is the
    // cheaper one (measured, tools/r04_consol2.sh: 4.0 against 4.6 ms for 1 024 fields of 512^2; 9.7 against 2.1 ms for 32 fields of
    // 2 048^2; 1.27 against 0.50 ms for 64 fields of 512^2).  FSQ_CONSOLIDATE_BLOCKS / FSQ_CONSOLIDATE_COMPONENTS force one.
    const int block_threads = ((int64_t)H * W > (1 << 20)) ? 1024 : 512;
    bool blocks = (long long)n_fields * (block_threads / 64) >= 8192;
    if (getenv("FSQ_CONSOLIDATE_BLOCKS")) blocks = true;
    if (getenv("FSQ_CONSOLIDATE_COMPONENTS")) blocks = false;
    if (blocks) {
        // round 4's first form: one block of 8 / 16 waves per field taking the turns by dependency (see k5_consolidate)
        int* grid = ws;
        unsigned char* turn = (unsigned char*)(ws + (size_t)n_fields * px);
        int* chunks_b = ws + (size_t)n_fields * 2 * px;
        FSQ_HIP_CHECK(hipMemsetAsync(grid, 0xFF, (size_t)n_fields * px * 4, s));
        hipLaunchKernelGGL(k5_consolidate, dim3(n_fields), dim3(block_threads), 0, s, d_rows, d_counts, d_offsets, H, W, r2_threshold,
                           radius, py2_round, grid, turn, chunks_b, (long long)k5_chunk_stride(H, W), d_keep, d_nkeep);
    } else {
        // (only the grids have to be -1: one strided memset over the fields' [grid | parent | last] blocks)
        FSQ_HIP_CHECK(hipMemset2DAsync(ws, (size_t)stride * 4, 0xFF, px * 4, n_fields, s));
        // blocks (of one wave) per field: enough waves to fill the chip whatever the number of fields, each taking 64 candidates at a time
        const int bpf = (int)std::max<long long>(8, std::min<long long>(((long long)px + 63) / 64, 16384 / n_fields));
        const dim3 g(bpf, n_fields);
        hipLaunchKernelGGL(k5c_insert, g, dim3(64), 0, s, d_rows, d_counts, d_offsets, H, W, r2_threshold, ws, stride);
        hipLaunchKernelGGL(k5c_union, g, dim3(64), 0, s, d_rows, d_counts, d_offsets, H, W, radius, ws, stride);
        hipLaunchKernelGGL(k5c_flatten, g, dim3(64), 0, s, d_counts, d_offsets, H, W, ws, stride);
        hipLaunchKernelGGL(k5c_turns, g, dim3(64), 0, s, d_rows, d_counts, d_offsets, H, W, radius, ws, stride);
        hipLaunchKernelGGL(k5c_finish, dim3(n_fields), dim3(512), 0, s, d_rows, d_counts, d_offsets, H, W, py2_round, ws, stride, chunks,
                           (long long)k5_chunk_stride(H, W), d_keep, d_nkeep);
    }
    hipLaunchKernelGGL(k5_total, dim3(1), dim3(1), 0, s, d_nkeep, n_fields);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}
