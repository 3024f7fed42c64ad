// fsq_consolidate.hip - K5: R^2 filter, consolidation of competing PSFs and re-keying, per field.
// Reference: pflib.find_peptides, pflib.py:466 (filter), 477-512 (consolidation), 514-519 (re-key).
//
// The reference's loops are sequential and order dependent (raster order of the candidate pixels,
// raster order inside each search window).  One 64-lane wave owns one field and walks the surviving
// candidates in that order; the window scan of one candidate (up to (2r+5)^2 cells) is done by the
// lanes in parallel and resolved with ballots so that the outcome equals the sequential scan:
//   rivals are visited in raster order; every rival with a smaller R^2 is deleted until the first
//   rival whose R^2 is not smaller - then the candidate itself is deleted and the scan stops.
// A pixel->row grid (int32 per pixel, workspace) plays the role of the reference's dict.
#include "fsq_common.h"
#include "fsq_devmath.h"

namespace {

__device__ __forceinline__ double round_key(double x, int py2) { return py2 ? round(x) : rint(x); }

// grid = n_fields blocks of 64 threads (one wave per field)
__global__ void __launch_bounds__(64) k5_consolidate(FsqRow* __restrict__ rows, const int* __restrict__ counts,
                                                     const int* __restrict__ offsets, int H, int W, double r2_thr,
                                                     int radius, int py2, int* __restrict__ grid_all,
                                                     int* __restrict__ keep, int* __restrict__ nkeep)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const int off = offsets[f], cnt = counts[f];
    int* grid = grid_all + (size_t)f * H * W;          // pre-set to -1
    FsqRow* R = rows + off;
    const double rr = (double)(radius * radius);

    // dict insertion (setdefault, pflib.py:477): survivors of the R^2 filter (NaN passes, :466).  Their indices are
    // also compacted, in order, into this field's slice of `keep` (scratch until the final list is written): the two
    // sequential walks below then touch ~15 % of the candidates and read each survivor's row ONCE, 64 rows at a time,
    // instead of chasing two dependent global loads per candidate.
    int* surv = keep + off;
    int nsurv = 0;
    for (int base = 0; base < cnt; base += 64) {
        const int i = base + lane;
        bool ok = false;
        if (i < cnt) {
            R[i].key_h = -1; R[i].key_w = -1;
            ok = !(R[i].r2 < r2_thr);
            if (ok) grid[(size_t)R[i].h * W + R[i].w] = i;
        }
        const unsigned long long m = __ballot(ok);
        if (ok) surv[nsurv + __popcll(m & ((1ull << lane) - 1ull))] = i;
        nsurv += __popcll(m);
    }
    __syncthreads();

    // consolidation, pflib.py:479-512
    for (int sbase = 0; sbase < nsurv; sbase += 64) {
        const int nb = min(64, nsurv - sbase);
        int my_i = -1, my_h = 0, my_w = 0;
        double my_h0 = 0., my_w0 = 0., my_r2 = 0.;
        if (lane < nb) {
            my_i = surv[sbase + lane];
            my_h = R[my_i].h; my_w = R[my_i].w; my_h0 = R[my_i].h0; my_w0 = R[my_i].w0; my_r2 = R[my_i].r2;
        }
        for (int j = 0; j < nb; j++) {
            const int i = __shfl(my_i, j), h = __shfl(my_h, j), w = __shfl(my_w, j);
            if (grid[(size_t)h * W + w] != i) continue;        // already deleted by an earlier candidate (wave-uniform)
            const double h0 = __shfl(my_h0, j), w0 = __shfl(my_w0, j), r2i = __shfl(my_r2, j);
            const int h_lo = max(0, h - radius - 2), h_hi = min(h + radius + 3, H);
            const int w_lo = max(0, w - radius - 2), w_hi = min(w + radius + 3, W);
            const int ww = w_hi - w_lo, ncell = (h_hi - h_lo) * ww;
            bool dead = false;
            for (int base = 0; base < ncell && !dead; base += 64) {
                int c = base + lane;
                bool rival = false, lose = false;
                size_t cell = 0;
                if (c < ncell) {
                    int hd = h_lo + c / ww, wd = w_lo + c % ww;
                    cell = (size_t)hd * W + wd;
                    int k = grid[cell];
                    if (k >= 0 && !(hd == h && wd == w)) {
                        double dh = h0 - R[k].h0, dw = w0 - R[k].w0;
                        if (!(fsq_pow2(dh) + fsq_pow2(dw) > rr)) {       // numpy scalar **2, pflib.py:505
                            rival = true;
                            lose = !(r2i > R[k].r2);                    // pflib.py:508
                        }
                    }
                }
                unsigned long long mlose = __ballot(lose);
                int first = mlose ? (__ffsll((long long)mlose) - 1) : 64;
                if (rival && lane < first) grid[cell] = -1;             // rivals with smaller R^2 die
                if (mlose) {
                    if (lane == 0) grid[(size_t)h * W + w] = -1;        // the candidate itself dies, scan stops
                    dead = true;
                }
                __syncthreads();
            }
        }
    }

    // re-key, pflib.py:514-519.  Whether an entry is still alive cannot change during this loop (entries only ever
    // move into EMPTY cells), so liveness, the rounded key and the key write are done 64 survivors at a time; only
    // the entries whose key actually moves are then replayed one by one in index order, because the reference's
    // assert looks at the dict as it is at that moment.
    __shared__ int s_assert;
    if (lane == 0) s_assert = 0;
    __syncthreads();
    for (int sbase = 0; sbase < nsurv; sbase += 64) {
        const int s_ = sbase + lane;
        bool moved = false;
        int i = -1, hr = 0, wr = 0;
        size_t g = 0;
        if (s_ < nsurv) {
            i = surv[s_];
            g = (size_t)R[i].h * W + R[i].w;
            if (grid[g] == i) {
                hr = (int)round_key(R[i].h0, py2); wr = (int)round_key(R[i].w0, py2);
                R[i].key_h = hr; R[i].key_w = wr;
                moved = (hr != R[i].h) || (wr != R[i].w);
            }
        }
        unsigned long long mm = __ballot(moved);
        while (mm) {
            const int b = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            if (lane == b) {
                grid[g] = -1;
                if (hr >= 0 && hr < H && wr >= 0 && wr < W) {
                    size_t g2 = (size_t)hr * W + wr;
                    if (grid[g2] >= 0) s_assert = 1;                 // assert (h_0_r, w_0_r) not in pixel_bins
                    else grid[g2] = i;
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();

    // kept list in the reference's dict order: untouched keys in insertion order, re-keyed ones appended
    int nk = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int base = 0; base < cnt; base += 64) {
            int i = base + lane;
            bool sel = false;
            if (i < cnt && R[i].key_h >= 0) {
                bool moved = (R[i].key_h != R[i].h) || (R[i].key_w != R[i].w);
                sel = (pass == 0) ? !moved : moved;
            }
            unsigned long long m = __ballot(sel);
            if (sel) keep[off + nk + __popcll(m & ((1ull << lane) - 1ull))] = off + i;
            nk += __popcll(m);
        }
    if (lane == 0) nkeep[f] = s_assert ? -1 : nk;
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

extern "C" int64_t fsq_consolidate_workspace_bytes(int n_fields, int H, int W)
{
    if (n_fields < 1 || H < 1 || W < 1) return FSQ_EINVAL;
    return (int64_t)n_fields * H * W * 4;
}

extern "C" int fsq_consolidate(FsqRow* d_rows, const int32_t* d_counts, const int32_t* d_offsets, int n_fields, int H,
                               int W, double r2_threshold, int radius, int py2_round, int32_t* d_keep, int32_t* d_nkeep,
                               void* d_workspace, int64_t workspace_bytes, void* stream)
{
    if (radius < 2) return FSQ_EINVAL;                                // pflib.py:431-432 -> ValueError
    if (n_fields < 1 || H < 5 || W < 5 || !d_rows || !d_counts || !d_offsets || !d_keep || !d_nkeep || !d_workspace) return FSQ_EINVAL;
    if (workspace_bytes < (int64_t)n_fields * H * W * 4) return FSQ_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    FSQ_HIP_CHECK(hipMemsetAsync(d_workspace, 0xFF, (size_t)n_fields * H * W * 4, s));
    hipLaunchKernelGGL(k5_consolidate, dim3(n_fields), dim3(64), 0, s, d_rows, d_counts, d_offsets, H, W, r2_threshold,
                       radius, py2_round, (int*)d_workspace, d_keep, d_nkeep);
    hipLaunchKernelGGL(k5_total, dim3(1), dim3(1), 0, s, d_nkeep, n_fields);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}
