/*
 * fsq_track_oracle.c - CPU restatement of the reference's greedy particle tracking.
 * TEST INFRASTRUCTURE (the checker): never linked into the product.
 *
 * Follows flexlibrary.py of the reference:
 *   Experiment.accumulate_offsets        flexlibrary.py:567-593   (python sum(): left to right from the int 0)
 *   Experiment.discard_dropouts          flexlibrary.py:626-678
 *   Experiment.greedy_particle_tracking  flexlibrary.py:680-1027
 * and, for the pair distance, scipy.spatial.distance.euclidean (flexlibrary.py:927) = scipy.linalg.norm = BLAS dnrm2
 * of the 2-vector of coordinate differences.  The OpenBLAS x86-64 dnrm2 kernel (kernel/x86_64/nrm2.S, pinned by
 * tests/test_tracking.py against the scipy of the build container) works in x87 extended precision:
 *   d = (double) sqrtl( (long double)dh*dh + (long double)dw*dw )        every operation rounded to 64 bits, then to 53
 * which differs from sqrt(dh*dh + dw*dw) in double for ~16 % of the 1/20-pixel displacement vectors.
 * The dense per-frame object arrays of the reference (one dict per pixel) are int grids here; iteration orders
 * (numpy.ndenumerate = raster) and the stable sort of the candidate pairs are kept.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fsq_oracle.h"

double fsq_o_euclid2(double dh, double dw)
{
    volatile long double a = (long double)dh * (long double)dh;     /* fmul: rounded to 64 bits */
    volatile long double b = (long double)dw * (long double)dw;
    volatile long double s = a + b;                                  /* faddp */
    volatile long double r = sqrtl(s);                               /* fsqrt */
    return (double)r;                                                /* fstpl */
}

/* Python 2 round(): half away from zero (flexlibrary.py:850, 880) */
static long py2_round(double x) { return (long)(x >= 0 ? floor(x + 0.5) : ceil(x - 0.5)); }

typedef struct { double dist; int a_cell, d_cell, a_spot, d_spot; } Pair;

static int pair_cmp(const void* pa, const void* pb)
{
    const Pair* a = (const Pair*)pa; const Pair* b = (const Pair*)pb;
    if (a->dist < b->dist) return -1;
    if (a->dist > b->dist) return 1;
    /* python's sorted() is stable: ties keep generation order = ancestors in raster order of the cache,
       each one's descendants in raster order of its window (flexlibrary.py:896-935) */
    if (a->a_cell != b->a_cell) return a->a_cell < b->a_cell ? -1 : 1;
    return a->d_cell < b->d_cell ? -1 : (a->d_cell > b->d_cell);
}

/*
 * One field.  Spots of frame f are hw[start[f] .. start[f] + counts[f]) (Spot.h, Spot.w: integers), offsets[f] is
 * frame f's offset relative to frame f - 1 (offsets[0] must be (0, 0): ValueError, flexlibrary.py:581-583).
 * Outputs (all indexed by global spot number, -1 = none): prev / next links, kept flags; traces[n_traces][n_frames]
 * in the reference's order (heads by frame, then raster order of their bin), *n_discarded.
 * Returns 0; -1 ValueError; -2 AssertionError (two spots of a frame round to the same bin, flexlibrary.py:851-856);
 * -3 capacity of traces exceeded.
 */
int fsq_o_greedy_tracking(int n_frames, const int32_t* counts, const int32_t* hw, const double* offsets, int H, int W,
                          int candidate_radius, double spot_radius, int32_t* link_prev, int32_t* link_next, uint8_t* kept,
                          int32_t* traces, int64_t traces_cap, int32_t* n_traces, int32_t* n_discarded)
{
    if (n_frames < 1 || H < 1 || W < 1) return -1;
    if (offsets[0] != 0.0 || offsets[1] != 0.0) return -1;
    int* start = (int*)malloc(sizeof(int) * (n_frames + 1));
    start[0] = 0;
    for (int f = 0; f < n_frames; f++) start[f + 1] = start[f] + counts[f];
    const int total = start[n_frames];
    double* cum = (double*)malloc(sizeof(double) * 2 * n_frames);
    {   /* accumulate_offsets: sum([o[0] for o in offsets[:f+1]]) - a fresh left-to-right sum per frame */
        for (int f = 0; f < n_frames; f++) {
            double sh = 0.0, sw = 0.0;
            for (int g = 0; g <= f; g++) { sh = sh + offsets[2 * g]; sw = sw + offsets[2 * g + 1]; }
            cum[2 * f] = sh; cum[2 * f + 1] = sw;
        }
    }
    int rc = 0;
    int disc = 0;
    int* frame_of = (int*)malloc(sizeof(int) * (total + 1));
    int* cell_of = (int*)malloc(sizeof(int) * (total + 1));
    for (int f = 0; f < n_frames; f++)
        for (int i = start[f]; i < start[f + 1]; i++) {
            frame_of[i] = f; link_prev[i] = -1; link_next[i] = -1;
            /* discard_dropouts (flexlibrary.py:657-677): position in frame 0's coordinates, then in every frame's */
            const double oh = hw[2 * i] + cum[2 * f], ow = hw[2 * i + 1] + cum[2 * f + 1];
            int ok = 1;
            for (int g = 0; g < n_frames && ok; g++) {
                const double gh = oh - cum[2 * g], gw = ow - cum[2 * g + 1];
                if (!(spot_radius <= gh && gh < H - 0.5 - spot_radius && spot_radius <= gw && gw < W - 0.5 - spot_radius)) ok = 0;
            }
            kept[i] = (uint8_t)ok;
            if (!ok) disc++;
        }
    /* frame_bins: per frame an H x W grid of spot numbers */
    int** bins = (int**)malloc(sizeof(int*) * n_frames);
    for (int f = 0; f < n_frames; f++) {
        bins[f] = (int*)malloc(sizeof(int) * (size_t)H * W);
        for (long k = 0; k < (long)H * W; k++) bins[f][k] = -1;
    }
    for (int f = 0; f < n_frames && rc == 0; f++)
        for (int i = start[f]; i < start[f + 1]; i++) {
            if (!kept[i]) continue;
            const double h = hw[2 * i] + cum[2 * f], w = hw[2 * i + 1] + cum[2 * f + 1];
            const long rh = py2_round(h), rw = py2_round(w);            /* inside the frame: discard_dropouts saw g = 0 */
            if (bins[f][rh * W + rw] != -1) { rc = -2; break; }
            bins[f][rh * W + rw] = i;
            cell_of[i] = (int)(rh * W + rw);
        }
    int* cache = (int*)malloc(sizeof(int) * (size_t)H * W);              /* ancestor_cache */
    for (long k = 0; k < (long)H * W; k++) cache[k] = -1;
    Pair* pairs = NULL; size_t pcap = 0;
    const int r = candidate_radius;
    for (int f = 1; f < n_frames && rc == 0; f++) {
        /* merge the spots of frame f - 1 into the cache (an older entry of the same bin is replaced, :884-897) */
        for (long k = 0; k < (long)H * W; k++)
            if (bins[f - 1][k] != -1) cache[k] = bins[f - 1][k];
        size_t np = 0;
        for (int ah = 0; ah < H; ah++)
            for (int aw = 0; aw < W; aw++) {
                const int a = cache[ah * W + aw];
                if (a == -1) continue;
                const int af = frame_of[a];
                int h0 = ah - r - 2; if (h0 < 0) h0 = 0;
                int w0 = aw - r - 2; if (w0 < 0) w0 = 0;
                int h1 = ah + r + 3; if (h1 > H) h1 = H;
                int w1 = aw + r + 3; if (w1 > W) w1 = W;
                const double a_h = hw[2 * a] + cum[2 * af], a_w = hw[2 * a + 1] + cum[2 * af + 1];
                for (int dh = h0; dh < h1; dh++)
                    for (int dw = w0; dw < w1; dw++) {
                        const int d = bins[f][dh * W + dw];
                        if (d == -1) continue;
                        const double d_h = hw[2 * d] + cum[2 * f], d_w = hw[2 * d + 1] + cum[2 * f + 1];
                        const double dist = fsq_o_euclid2(a_h - d_h, a_w - d_w);
                        if (dist < (double)r) {
                            if (np == pcap) { pcap = pcap ? 2 * pcap : 1024; pairs = (Pair*)realloc(pairs, pcap * sizeof(Pair)); }
                            pairs[np].dist = dist; pairs[np].a_cell = ah * W + aw; pairs[np].d_cell = dh * W + dw;
                            pairs[np].a_spot = a; pairs[np].d_spot = d;
                            np++;
                        }
                    }
            }
        if (np > 1) qsort(pairs, np, sizeof(Pair), pair_cmp);             /* (keys are unique: a total order) */
        for (size_t k = 0; k < np; k++) {
            const Pair* p = &pairs[k];
            if (cache[p->a_cell] == -1) continue;                         /* ancestor has been paired */
            if (link_prev[p->d_spot] != -1) continue;                     /* descendant has been paired */
            link_prev[p->d_spot] = p->a_spot;
            link_next[p->a_spot] = p->d_spot;
            cache[p->a_cell] = -1;
        }
    }
    /* traces: heads (no ancestor) by frame, raster order of the bins (:975-1026) */
    int nt = 0;
    for (int f = 0; f < n_frames && rc == 0; f++)
        for (long k = 0; k < (long)H * W && rc == 0; k++) {
            const int s = bins[f][k];
            if (s == -1 || link_prev[s] != -1) continue;
            if (nt >= traces_cap) { rc = -3; break; }
            int32_t* t = traces + (size_t)nt * n_frames;
            for (int g = 0; g < n_frames; g++) t[g] = -1;
            for (int c = s; c != -1; c = link_next[c]) t[frame_of[c]] = c;
            nt++;
        }
    *n_traces = nt; *n_discarded = disc;
    for (int f = 0; f < n_frames; f++) free(bins[f]);
    free(bins); free(cache); free(pairs); free(frame_of); free(cell_of); free(cum); free(start);
    return rc;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Experiment.luminosity_centroid_particle_tracking / next_frame_spot_by_luminosity_centroid,
 * flexlibrary.py:1173-1317, for Spots of size 5 and integer offsets (a non-integer offset makes the reference's
 * image slicing raise TypeError under the numpy of the build container).
 * frames uint16[F][H][W]; init_hw int32[n][2]; offsets int64[F][2] (offsets[f] is applied between frame f-1 and f;
 * NULL = all zero).  out_hw int32[n][F][2], present uint8[n][F] (0 = None).
 * Returns 0, or -1 for the reference's ValueError (all-zero search window: centre of mass is NaN, int(round(nan))).
 */
static long slice_bound(long v, long n) { if (v < 0) { v += n; if (v < 0) v = 0; } else if (v > n) v = n; return v; }

static int spot_fits(long h, long w, int size, int H, int W)               /* Spot.__init__, flexlibrary.py:100-111 */
{
    const long r = (size - 1) / 2;
    return 0 <= h - r && h + r < H && 0 <= w - r && w + r < W;
}

/* pixel k of a stack stored as uint16 (wide = 0) or uint32 (wide = 1, values < 2^31) */
static inline uint64_t trk_px(const void* frames, int wide, size_t k)
{
    return wide ? (uint64_t)((const uint32_t*)frames)[k] : (uint64_t)((const uint16_t*)frames)[k];
}

static int centroid_tracking_any(const void* frames, int wide, int F, int H, int W, const int32_t* init_hw, int n, int size,
                                 int search_radius, double s_n_cutoff, const int64_t* offsets, int32_t* out_hw, uint8_t* present)
{
    if (size != 5 || search_radius < 0) return -2;
    const int R = search_radius, D = 2 * R + 1;
    for (int i = 0; i < n; i++) {
        long ph = init_hw[2 * i], pw = init_hw[2 * i + 1];                   /* prior_frame_spot */
        out_hw[((size_t)i * F) * 2] = (int32_t)ph; out_hw[((size_t)i * F) * 2 + 1] = (int32_t)pw;
        present[(size_t)i * F] = 1;
        for (int f = 1; f < F; f++) {
            const size_t img0 = (size_t)f * H * W;                        /* frame f starts at pixel img0 */
            const long oh = ph - (offsets ? offsets[2 * f] : 0), ow = pw - (offsets ? offsets[2 * f + 1] : 0);
            /* numpy slice image[oh-R : oh+R+1, ow-R : ow+R+1] (negative bounds wrap, flexlibrary.py:1223-1226) */
            const long h0 = slice_bound(oh - R, H), h1 = slice_bound(oh + R + 1, H);
            const long w0 = slice_bound(ow - R, W), w1 = slice_bound(ow + R + 1, W);
            int found = 0; long nh = 0, nw = 0;
            if (h1 - h0 == D && w1 - w0 == D) {
                /* scipy.ndimage.center_of_mass on the int64 window: integer sums (exact), then one true division per axis */
                uint64_t norm = 0, sh = 0, sw = 0;
                for (int a = 0; a < D; a++)
                    for (int b = 0; b < D; b++) {
                        const uint64_t v = trk_px(frames, wide, img0 + (size_t)(h0 + a) * W + (w0 + b));
                        norm += v; sh += v * (uint64_t)a; sw += v * (uint64_t)b;
                    }
                if (norm == 0) return -1;
                const double ch = (double)sh / (double)norm, cw = (double)sw / (double)norm;
                const long rh = py2_round((ch + (double)oh) - (double)R), rw = py2_round((cw + (double)ow) - (double)R);
                if (spot_fits(rh, rw, size, H, W)) {
                    int64_t roi[25];
                    for (int a = 0; a < 5; a++)
                        for (int b = 0; b < 5; b++) roi[a * 5 + b] = (int64_t)trk_px(frames, wide, img0 + (size_t)(rh - 2 + a) * W + (rw - 2 + b));
                    found = 1; nh = rh; nw = rw;
                    if (fsq_o_illumina_s_n(roi) < s_n_cutoff) {               /* :1248-1259: same coordinates as the prior spot */
                        if (spot_fits(ph, pw, size, H, W)) { nh = ph; nw = pw; } else found = 0;
                    }
                }
            }
            present[(size_t)i * F + f] = (uint8_t)found;
            out_hw[((size_t)i * F + f) * 2] = found ? (int32_t)nh : -1;
            out_hw[((size_t)i * F + f) * 2 + 1] = found ? (int32_t)nw : -1;
            if (found) { ph = nh; pw = nw; }
        }
    }
    return 0;
}

int fsq_o_centroid_tracking(const uint16_t* frames, int F, int H, int W, const int32_t* init_hw, int n, int size,
                            int search_radius, double s_n_cutoff, const int64_t* offsets, int32_t* out_hw, uint8_t* present)
{
    return centroid_tracking_any(frames, 0, F, H, W, init_hw, n, size, search_radius, s_n_cutoff, offsets, out_hw, present);
}

/* the same on uint32 frames (values < 2^31) */
int fsq_o_centroid_tracking_u32(const uint32_t* frames, int F, int H, int W, const int32_t* init_hw, int n, int size,
                                int search_radius, double s_n_cutoff, const int64_t* offsets, int32_t* out_hw, uint8_t* present)
{
    return centroid_tracking_any(frames, 1, F, H, W, init_hw, n, size, search_radius, s_n_cutoff, offsets, out_hw, present);
}
