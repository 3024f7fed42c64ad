// fsq_track.hip - K8: greedy particle tracking of the peak tables across the frames of a field.
//
// Reference: Experiment.accumulate_offsets / discard_dropouts / greedy_particle_tracking, flexlibrary.py:567-1027.
// The reference keeps one Python dict per pixel per frame and walks whole frames with numpy.ndenumerate (O(frames x H x W)
// interpreter steps per field); here one 256-thread block owns a field and walks its frames in order:
//   spots -> bins (py2-rounded position in frame-0 coordinates), duplicate-bin check (the reference's assert, :851)
//   per frame f >= 1:  merge frame f-1 into the ancestor cache (a newer spot replaces an older one in the same bin),
//                      every live ancestor scans its (2r+5)^2 window of frame f's bins, pairs closer than r are kept,
//                      the pairs are ordered by (distance, ancestor bin, descendant bin) - python's stable sorted() over
//                      the reference's generation order - and accepted greedily
//   traces: heads (spots without ancestor) by frame and bin, followed along their descendant links.
// The pair distance is scipy's euclidean = OpenBLAS dnrm2 in x87 extended precision: fsq_x87.h.
// Per field in HBM: two int32 grids (bin -> spot of the current frame, bin -> cached ancestor) and the candidate-pair
// list of the current frame; the paired flags and the offsets live in LDS.
#include "fsq_common.h"
#include "fsq_x87.h"

namespace {

constexpr int TRK_SMAX = 32768;         // up to this many spots per field (all frames) the "has been paired" flags are bitmaps in LDS;
                                        // larger fields read the links themselves (round 4: no spot limit)
constexpr int TRK_FLDS = 64;            // frame tables (cumulative offsets, first spot of every frame) of up to this many frames live
                                        // in LDS; longer time series keep them in the caller's workspace (round 4: no frame limit)

__device__ __forceinline__ long trk_py2_round(double x) { return (long)(x >= 0 ? __builtin_floor(x + 0.5) : __builtin_ceil(x - 0.5)); }

struct TrkPairs {                        // candidate pairs of the current frame (HBM, pair_cap entries each)
    double* dist;
    unsigned long long* key2;            // ancestor bin << 32 | descendant bin
    int *a_spot, *d_spot, *order;
};
__host__ __device__ inline size_t trk_pairs_bytes(long long pair_cap) { return (size_t)pair_cap * (8 + 8 + 4 + 4 + 4); }

struct TrkShared {
    unsigned a_taken[TRK_SMAX / 32], d_taken[TRK_SMAX / 32];
    double cum[TRK_FLDS][2];
    int fs[TRK_FLDS + 1];
    int hc[TRK_FLDS];                   // heads (trace starts) per frame
    int np, ndisc, status, nheads;
};

__global__ void __launch_bounds__(256) k8_track(const int32_t* __restrict__ hw, const int32_t* __restrict__ field_start,
                                                const int32_t* __restrict__ counts, const double* __restrict__ offsets,
                                                int F, int H, int W, int radius, double spot_radius,
                                                int32_t* __restrict__ prev, int32_t* __restrict__ next, uint8_t* __restrict__ kept,
                                                int32_t* __restrict__ traces, int32_t* __restrict__ n_traces,
                                                int32_t* __restrict__ n_disc, int32_t* __restrict__ status_out,
                                                int32_t* __restrict__ grids, unsigned char* __restrict__ pair_ws, int pair_cap,
                                                double* __restrict__ frame_ws)
{
    __shared__ TrkShared S;
    const int fld = blockIdx.x, t = threadIdx.x;
    const int s0 = field_start[fld], n = field_start[fld + 1] - s0;
    const size_t cells = (size_t)H * W;
    int32_t* bins = grids + (size_t)fld * 2 * cells;
    int32_t* cache = bins + cells;
    TrkPairs P;
    {
        unsigned char* base = pair_ws + (size_t)fld * trk_pairs_bytes(pair_cap);
        P.dist = (double*)base; P.key2 = (unsigned long long*)(base + (size_t)pair_cap * 8);
        P.a_spot = (int*)(base + (size_t)pair_cap * 16); P.d_spot = P.a_spot + pair_cap; P.order = P.d_spot + pair_cap;
    }
    const int32_t* my_hw = hw + (size_t)s0 * 2;
    int32_t* my_prev = prev + s0;
    int32_t* my_next = next + s0;
    uint8_t* my_kept = kept + s0;
    // frame tables: cum[f] = the offsets accumulated up to frame f, fs[f] = number of the first spot of frame f
    double* cum = (F <= TRK_FLDS) ? &S.cum[0][0] : frame_ws + (size_t)fld * ((size_t)F * 2 + (size_t)(F + 2) / 2 + 1 + (size_t)(F + 1) / 2);
    int* fs = (F <= TRK_FLDS) ? S.fs : (int*)(cum + (size_t)F * 2);
    int* hc = (F <= TRK_FLDS) ? S.hc : (int*)(cum + (size_t)F * 2 + (size_t)(F + 2) / 2 + 1);
    const bool lds_flags = n <= TRK_SMAX;
    const double* off = offsets + (size_t)fld * F * 2;
    if (t == 0) {
        S.np = 0; S.ndisc = 0; S.status = 0; S.nheads = 0;
        fs[0] = 0;
        for (int f = 0; f < F; f++) fs[f + 1] = fs[f] + counts[(size_t)fld * F + f];
        if (off[0] != 0.0 || off[1] != 0.0) S.status = FSQ_EINVAL;          // ValueError, flexlibrary.py:581-583
        if (fs[F] != n) S.status = FSQ_EINVAL;
    }
    for (int f = t; f < F; f += 256) {                  // accumulate_offsets: a fresh left-to-right sum per frame (flexlibrary.py:585-600)
        double sh = 0.0, sw = 0.0;
        for (int g = 0; g <= f; g++) { sh = sh + off[2 * g]; sw = sw + off[2 * g + 1]; }
        cum[2 * f] = sh; cum[2 * f + 1] = sw;
    }
    for (int k = t; k < TRK_SMAX / 32; k += 256) { S.a_taken[k] = 0; S.d_taken[k] = 0; }
    __syncthreads();
    if (S.status != 0) {
        if (t == 0) { status_out[fld] = S.status; n_traces[fld] = 0; n_disc[fld] = 0; }
        return;
    }
    auto frame_of = [&](int i) {                        // the frame whose spots include number i (fs is non-decreasing)
        int lo = 0, hi = F - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (i >= fs[mid + 1]) lo = mid + 1; else hi = mid; }
        return lo;
    };
    auto pos = [&](int i, int f, double* h, double* w) { *h = my_hw[2 * i] + cum[2 * f]; *w = my_hw[2 * i + 1] + cum[2 * f + 1]; };
    auto cell_of = [&](int i, int f) {
        double h, w; pos(i, f, &h, &w);
        return (int)(trk_py2_round(h) * W + trk_py2_round(w));
    };
    // ---- discard_dropouts (flexlibrary.py:657-677) ------------------------------------------------------------
    for (int i = t; i < n; i += 256) {
        const int f = frame_of(i);
        double oh, ow; pos(i, f, &oh, &ow);
        bool ok = true;
        for (int g = 0; g < F && ok; g++) {
            const double gh = oh - cum[2 * g], gw = ow - cum[2 * g + 1];
            ok = (spot_radius <= gh && gh < H - 0.5 - spot_radius && spot_radius <= gw && gw < W - 0.5 - spot_radius);
        }
        my_kept[i] = ok ? 1 : 0;
        my_prev[i] = -1; my_next[i] = -1;
        if (!ok) atomicAdd(&S.ndisc, 1);
    }
    __syncthreads();
    // ---- two spots of one frame in one bin: the reference's assert (flexlibrary.py:851-856) ---------------------
    for (int f = 0; f < F; f++) {
        for (int i = fs[f] + t; i < fs[f + 1]; i += 256)
            if (my_kept[i] && atomicCAS(&bins[cell_of(i, f)], -1, i) != -1) S.status = FSQ_EASSERT;
        __syncthreads();
        for (int i = fs[f] + t; i < fs[f + 1]; i += 256)
            if (my_kept[i]) bins[cell_of(i, f)] = -1;
        __syncthreads();
    }
    if (S.status != 0) {
        if (t == 0) { status_out[fld] = S.status; n_traces[fld] = 0; n_disc[fld] = S.ndisc; }
        return;
    }
    // ---- main loop (flexlibrary.py:858-971) ---------------------------------------------------------------------
    for (int f = 1; f < F; f++) {
        for (int i = fs[f - 1] + t; i < fs[f]; i += 256)           // merge frame f - 1 into the ancestor cache
            if (my_kept[i]) cache[cell_of(i, f - 1)] = i;
        for (int i = fs[f] + t; i < fs[f + 1]; i += 256)           // bins of frame f
            if (my_kept[i]) bins[cell_of(i, f)] = i;
        if (t == 0) S.np = 0;
        __syncthreads();
        for (int a = t; a < fs[f]; a += 256) {                        // every live ancestor looks at its window
            if (!my_kept[a]) continue;
            const int af = frame_of(a);
            double a_h, a_w; pos(a, af, &a_h, &a_w);
            const int ah = (int)trk_py2_round(a_h), aw = (int)trk_py2_round(a_w);
            if (cache[(size_t)ah * W + aw] != a) continue;             // paired earlier, or replaced by a newer spot
            const int h0 = max(ah - radius - 2, 0), h1 = min(ah + radius + 3, H);
            const int w0 = max(aw - radius - 2, 0), w1 = min(aw + radius + 3, W);
            for (int dh = h0; dh < h1; dh++)
                for (int dw = w0; dw < w1; dw++) {
                    const int d = bins[(size_t)dh * W + dw];
                    if (d == -1) continue;
                    double d_h, d_w; pos(d, f, &d_h, &d_w);
                    const double dist = fsq_dnrm2_2(a_h - d_h, a_w - d_w);
                    if (dist < (double)radius) {
                        const int at = atomicAdd(&S.np, 1);
                        if (at < pair_cap) {
                            P.dist[at] = dist;
                            P.key2[at] = ((unsigned long long)(unsigned)(ah * W + aw) << 32) | (unsigned)(dh * W + dw);
                            P.a_spot[at] = a; P.d_spot[at] = d;
                        }
                    }
                }
        }
        __syncthreads();
        const int np = S.np;
        if (np > pair_cap) {
            if (t == 0) { status_out[fld] = FSQ_ERANGE; n_traces[fld] = 0; n_disc[fld] = S.ndisc; }
            return;                                                     // (uniform: every thread sees the same np)
        }
        for (int i = t; i < np; i += 256) {                             // rank = pairs that sort before this one
            const double di = P.dist[i];
            const unsigned long long ki = P.key2[i];
            int r = 0;
            for (int j = 0; j < np; j++) {
                const double dj = P.dist[j];
                r += (dj < di) || (dj == di && P.key2[j] < ki);
            }
            P.order[r] = i;
        }
        __syncthreads();
        if (t == 0) {                                                   // greedy acceptance in sorted order
            for (int k = 0; k < np; k++) {
                const int i = P.order[k], a = P.a_spot[i], d = P.d_spot[i];
                if (lds_flags) {
                    if ((S.a_taken[a >> 5] >> (a & 31)) & 1u) continue;     // ancestor has been paired
                    if ((S.d_taken[d >> 5] >> (d & 31)) & 1u) continue;     // descendant has been paired
                    S.a_taken[a >> 5] |= 1u << (a & 31);
                    S.d_taken[d >> 5] |= 1u << (d & 31);
                } else if (my_next[a] != -1 || my_prev[d] != -1) continue;   // (the same two facts, read off the links this thread writes)
                my_prev[d] = a; my_next[a] = d;
                cache[(unsigned)(P.key2[i] >> 32)] = -1;
            }
        }
        for (int i = fs[f] + t; i < fs[f + 1]; i += 256)           // bins back to empty
            if (my_kept[i]) bins[cell_of(i, f)] = -1;
        __syncthreads();
    }
    for (int i = t; i < n; i += 256)                                    // leave the cache grid empty for the next call
        if (my_kept[i]) cache[cell_of(i, frame_of(i))] = -1;
    // ---- traces: heads by (frame, bin), each followed along its links (flexlibrary.py:975-1026) -------------------
    int32_t* my_traces = traces + (size_t)s0 * F;
    // a head's row = the heads of earlier frames + the heads of its own frame in smaller bins
    for (int f = t; f < F; f += 256) hc[f] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 256)
        if (my_kept[i] && my_prev[i] == -1) atomicAdd(&hc[frame_of(i)], 1);
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int f = 0; f < F; f++) { const int c_ = hc[f]; hc[f] = run; run += c_; }
        S.nheads = run;
    }
    __syncthreads();
    for (int i = t; i < n; i += 256) {
        if (!my_kept[i] || my_prev[i] != -1) continue;
        const int f = frame_of(i), cell = cell_of(i, f);
        int r = hc[f];
        for (int j = fs[f]; j < fs[f + 1]; j++)
            r += (my_kept[j] && my_prev[j] == -1 && cell_of(j, f) < cell);
        int32_t* row = my_traces + (size_t)r * F;
        for (int g = 0; g < F; g++) row[g] = -1;
        for (int c = i; c != -1; c = my_next[c]) row[frame_of(c)] = c;
    }
    __syncthreads();
    if (t == 0) { status_out[fld] = 0; n_traces[fld] = S.nheads; n_disc[fld] = S.ndisc; }
}

// ---------------------------------------------------------------------------------------------------------------------
// K9: luminosity-centroid tracking - Experiment.luminosity_centroid_particle_tracking with
// next_frame_spot_by_luminosity_centroid, flexlibrary.py:1173-1317.  One thread follows one initial spot through the
// frames of its field: the (2R+1)^2 window around the offset-corrected position of the last sighting (numpy slice
// semantics, negative bounds wrap), its centre of mass (exact integer sums, one fp64 division per axis), Python-2
// rounding, the Spot-fits-the-image test (flexlibrary.py:100-111), illumina_s_n of the 5x5 area (pflib.py:261-281, numpy's
// summation order for 16 values) against the cut-off; below it the spot keeps the coordinates of its last sighting.
__device__ __forceinline__ long trk_slice_bound(long v, long n) { if (v < 0) { v += n; if (v < 0) v = 0; } else if (v > n) v = n; return v; }
__device__ __forceinline__ bool trk_spot_fits(long h, long w, int H, int W) { return 0 <= h - 2 && h + 2 < H && 0 <= w - 2 && w + 2 < W; }

template <typename PX>
__device__ double trk_illumina_s_n(const PX* __restrict__ img, int W, long h, long w)
{
    double op[16];
    int t = 0;
    unsigned mx = 0;
    const PX* base = img + (size_t)(h - 2) * W + (w - 2);
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++) { const unsigned v = base[(size_t)a * W + b]; mx = v > mx ? v : mx; }
    for (int b = 0; b < 5; b++) op[t++] = (double)base[b];
    for (int b = 0; b < 5; b++) op[t++] = (double)base[(size_t)4 * W + b];
    for (int a = 1; a < 4; a++) { op[t++] = (double)base[(size_t)a * W]; op[t++] = (double)base[(size_t)a * W + 4]; }
    double isum = 0.0;
    for (int k = 0; k < 16; k++) isum += op[k];                     // integers: exact in any order
    const double mean = isum / 16.0;
    double rr[8];
    for (int k = 0; k < 8; k++) { const double d0 = op[k] - mean, d1 = op[8 + k] - mean; rr[k] = d0 * d0 + d1 * d1; }
    double res = ((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]));   // numpy pairwise sum, n = 16
    res = 0.0 + res;
    return ((double)mx - mean) / __builtin_sqrt(res / 16.0);
}

// PX: uint16_t, or uint32_t for FSQ_PIXELS_U32 frames (values < 2^31; the window sums stay far below 2^64 for any radius an image allows)
template <typename PX>
__global__ void __launch_bounds__(256) k9_centroid_track(const PX* __restrict__ frames, int n_fields, int F, int H, int W,
                                                         const int32_t* __restrict__ init_hw, const int32_t* __restrict__ spot_field,
                                                         long long n, int R, double s_n_cutoff, const long long* __restrict__ offsets,
                                                         int32_t* __restrict__ out_hw, uint8_t* __restrict__ present,
                                                         int32_t* __restrict__ n_errors)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int fld = spot_field[i];
    long ph = init_hw[2 * i], pw = init_hw[2 * i + 1];
    out_hw[(size_t)i * F * 2] = (int32_t)ph; out_hw[(size_t)i * F * 2 + 1] = (int32_t)pw;
    present[(size_t)i * F] = 1;
    const int D = 2 * R + 1;
    bool failed = false;
    for (int f = 1; f < F; f++) {
        const PX* img = frames + ((size_t)fld * F + f) * H * W;
        const long oh = ph - (offsets ? (long)offsets[((size_t)fld * F + f) * 2] : 0);
        const long ow = pw - (offsets ? (long)offsets[((size_t)fld * F + f) * 2 + 1] : 0);
        const long h0 = trk_slice_bound(oh - R, H), h1 = trk_slice_bound(oh + R + 1, H);
        const long w0 = trk_slice_bound(ow - R, W), w1 = trk_slice_bound(ow + R + 1, W);
        bool found = false;
        long nh = 0, nw = 0;
        if (!failed && h1 - h0 == D && w1 - w0 == D) {
            unsigned long long norm = 0, sh = 0, sw = 0;
            for (int a = 0; a < D; a++)
                for (int b = 0; b < D; b++) {
                    const unsigned long long v = img[(size_t)(h0 + a) * W + (w0 + b)];
                    norm += v; sh += v * (unsigned)a; sw += v * (unsigned)b;
                }
            if (norm == 0) failed = true;                               // NaN centroid: the reference raises ValueError
            else {
                const double ch = (double)sh / (double)norm, cw = (double)sw / (double)norm;
                const long rh = trk_py2_round((ch + (double)oh) - (double)R), rw = trk_py2_round((cw + (double)ow) - (double)R);
                if (trk_spot_fits(rh, rw, H, W)) {
                    found = true; nh = rh; nw = rw;
                    if (trk_illumina_s_n(img, W, rh, rw) < s_n_cutoff) {
                        if (trk_spot_fits(ph, pw, H, W)) { nh = ph; nw = pw; } else found = false;
                    }
                }
            }
        }
        present[(size_t)i * F + f] = found ? 1 : 0;
        out_hw[((size_t)i * F + f) * 2] = found ? (int32_t)nh : -1;
        out_hw[((size_t)i * F + f) * 2 + 1] = found ? (int32_t)nw : -1;
        if (found) { ph = nh; pw = nw; }
    }
    if (failed) atomicAdd(n_errors, 1);
}

__global__ void kx87check(const double* __restrict__ dh, const double* __restrict__ dw, long long n, double* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fsq_dnrm2_2(dh[i], dw[i]);
}

}  // namespace

// per field: two bin grids, the candidate-pair list of one frame, and - for time series of more than TRK_FLDS frames - the frame tables
static int64_t trk_frame_doubles(int n_frames) { return n_frames <= TRK_FLDS ? 0 : (int64_t)n_frames * 2 + ((int64_t)n_frames + 2) / 2 + 1 + ((int64_t)n_frames + 1) / 2; }
extern "C" int64_t fsq_track_workspace_bytes(int n_fields, int n_frames, int H, int W, int64_t pair_cap)
{
    if (n_fields < 1 || n_frames < 1 || H < 1 || W < 1 || pair_cap < 1) return FSQ_EINVAL;
    return (int64_t)n_fields * 2 * H * W * 4 + (((int64_t)n_fields * (int64_t)trk_pairs_bytes(pair_cap) + 7) & ~(int64_t)7) +
           (int64_t)n_fields * trk_frame_doubles(n_frames) * 8;
}

extern "C" int fsq_greedy_tracking(const int32_t* d_hw, const int32_t* d_field_start, const int32_t* d_counts,
                                   const double* d_offsets, int n_fields, int n_frames, int H, int W, int candidate_radius,
                                   double spot_radius, int32_t* d_prev, int32_t* d_next, uint8_t* d_kept, int32_t* d_traces,
                                   int32_t* d_n_traces, int32_t* d_n_discarded, int32_t* d_status, int64_t pair_cap,
                                   void* d_workspace, int64_t workspace_bytes, void* stream)
{
    if (n_fields < 1 || n_frames < 1 || H < 1 || W < 1 || candidate_radius < 0 || !(spot_radius >= 0)) return FSQ_EINVAL;
    if (!d_field_start || !d_counts || !d_offsets || !d_n_traces || !d_n_discarded || !d_status || !d_workspace) return FSQ_EINVAL;
    if ((long long)H * W >= (1ll << 31)) return FSQ_ENOTIMPL;
    if (pair_cap < 1 || pair_cap > 2000000000ll) return FSQ_EINVAL;
    if (workspace_bytes < fsq_track_workspace_bytes(n_fields, n_frames, H, W, pair_cap)) return FSQ_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    FSQ_HIP_CHECK(hipMemsetAsync(d_workspace, 0xFF, (size_t)n_fields * 2 * H * W * 4, s));
    unsigned char* pair_ws = (unsigned char*)d_workspace + (size_t)n_fields * 2 * H * W * 4;
    double* frame_ws = (double*)(pair_ws + (((size_t)n_fields * trk_pairs_bytes(pair_cap) + 7) & ~(size_t)7));
    hipLaunchKernelGGL(k8_track, dim3(n_fields), dim3(256), 0, s, d_hw, d_field_start, d_counts, d_offsets, n_frames, H, W,
                       candidate_radius, spot_radius, d_prev, d_next, d_kept, d_traces, d_n_traces, d_n_discarded, d_status,
                       (int32_t*)d_workspace, pair_ws, (int)pair_cap, frame_ws);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}

template <typename PX>
static int centroid_tracking_launch(const PX* d_frames, int n_fields, int n_frames, int H, int W, const int32_t* d_init_hw,
                                    const int32_t* d_spot_field, int64_t n, int search_radius, double s_n_cutoff,
                                    const int64_t* d_offsets, int32_t* d_out_hw, uint8_t* d_present, int32_t* d_n_errors, void* stream)
{
    if (n_fields < 1 || n_frames < 1 || H < 1 || W < 1 || n < 0 || search_radius < 0) return FSQ_EINVAL;
    if (!d_frames || !d_n_errors || (n > 0 && (!d_init_hw || !d_spot_field || !d_out_hw || !d_present))) return FSQ_EINVAL;
    if (sizeof(PX) == 4 && search_radius > 512) return FSQ_ENOTIMPL;           // (the weighted sums of a window - D^3 x 2^31 at most - must fit 63 bits, like scipy's int64 sums)
    hipStream_t s = (hipStream_t)stream;
    FSQ_HIP_CHECK(hipMemsetAsync(d_n_errors, 0, sizeof(int32_t), s));
    if (n > 0)
        hipLaunchKernelGGL(k9_centroid_track<PX>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_frames, n_fields, n_frames, H, W,
                           d_init_hw, d_spot_field, (long long)n, search_radius, s_n_cutoff, (const long long*)d_offsets, d_out_hw,
                           d_present, d_n_errors);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}

extern "C" int fsq_centroid_tracking(const uint16_t* d_frames, int n_fields, int n_frames, int H, int W, const int32_t* d_init_hw,
                                     const int32_t* d_spot_field, int64_t n, int search_radius, double s_n_cutoff,
                                     const int64_t* d_offsets, int32_t* d_out_hw, uint8_t* d_present, int32_t* d_n_errors,
                                     void* stream)
{
    return centroid_tracking_launch(d_frames, n_fields, n_frames, H, W, d_init_hw, d_spot_field, n, search_radius, s_n_cutoff,
                                    d_offsets, d_out_hw, d_present, d_n_errors, stream);
}

extern "C" int fsq_centroid_tracking_u32(const uint32_t* d_frames, int n_fields, int n_frames, int H, int W, const int32_t* d_init_hw,
                                         const int32_t* d_spot_field, int64_t n, int search_radius, double s_n_cutoff,
                                         const int64_t* d_offsets, int32_t* d_out_hw, uint8_t* d_present, int32_t* d_n_errors,
                                         void* stream)
{
    return centroid_tracking_launch(d_frames, n_fields, n_frames, H, W, d_init_hw, d_spot_field, n, search_radius, s_n_cutoff,
                                    d_offsets, d_out_hw, d_present, d_n_errors, stream);
}

extern "C" int fsq_selftest_dnrm2(const double* d_dh, const double* d_dw, int64_t n, double* d_out, void* stream)
{
    if (n < 0 || (n > 0 && (!d_dh || !d_dw || !d_out))) return FSQ_EINVAL;
    if (n > 0) hipLaunchKernelGGL(kx87check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_dh, d_dw, (long long)n, d_out);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}
