// fsq_detect.hip - K1/K2: candidate detection for a batch of fields.
// Reference: pflib._psf_candidates, pflib.py:217-258
//   int64 copy -> scipy.ndimage.median_filter(size, mode='reflect') -> img - min(med, img)
//   -> scipy.signal.correlate(., K, 'same') zero padded -> max(., 0)
//   -> thr = numpy.mean + c_std * numpy.std  (float64, numpy's summation order)
//   -> raster list of interior pixels with !(cm < thr)
//
// Data layout in HBM: images uint16[n_fields][H][W]; response cm int64[n_fields][H][W] (workspace);
// candidates int32[cap][3] = (field, h, w), fields in order, raster order inside a field.
//
// Kernels (all HBM-streaming; algorithmic bytes per field: 2 B/px read + 8 B/px written by K1,
// 8 B/px read by each of the two later passes):
//   k1_response      LDS-tiled fused median -> background subtraction -> correlation -> clamp, + sum(cm)
//   k2_sqdev_chunks  sum((cm-mean)^2) per 8192-pixel chunk in numpy's pairwise order (bit-exact std)
//   k2_threshold     per field: fold chunk sums left to right, thr = mean + c_std*sqrt(var)
//   k2_count / k2_scan_field / k2_scan_all / k2_write   ordered stream compaction (no atomics)
#include <algorithm>
#include <cstdlib>

#include "fsq_common.h"

namespace {
#include "fsq_median_net.h"

constexpr int TW = 64, TH = 16;          // output tile of k1 (1024 px, 256 threads x 4 px)
constexpr int MAXK = FSQ_MAX_KSIZE;      // largest correlation matrix / median window side
constexpr int CHUNK = 8192;              // numpy ufunc buffer size in elements (NPY_BUFSIZE)

struct DetectConst {
    int med, ksz;
    int K[MAXK * MAXK];
};

__device__ __forceinline__ int reflect_idx(int i, int n)
{   // scipy.ndimage 'reflect': (d c b a | a b c d | d c b a)
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return (i < n) ? i : p - 1 - i;
}

#define CE(a, b) { int lo_ = min(a, b); int hi_ = max(a, b); a = lo_; b = hi_; }

// after the call v[0] = min and v[S-1] = max of v[0..S-1]
template <int S>
__device__ __forceinline__ void minmax_ends(int* v)
{
#pragma unroll
    for (int i = 0; i < S / 2; i++) CE(v[i], v[S - 1 - i]);
#pragma unroll
    for (int i = 1; i < (S + 1) / 2; i++) CE(v[0], v[i]);
#pragma unroll
    for (int i = S / 2; i < S - 1; i++) CE(v[i], v[S - 1]);
}

template <int S>
struct Forget {
    // working set v[0..S-1]; drop its min and max, insert the next sample, recurse
    static __device__ __forceinline__ int run(int* v, const int* rest)
    {
        minmax_ends<S>(v);
        v[0] = rest[0];                  // min replaced by the next sample; max (v[S-1]) falls off the end
        return Forget<S - 1>::run(v, rest + 1);
    }
};
template <>
struct Forget<3> {
    static __device__ __forceinline__ int run(int* v, const int*)
    {
        minmax_ends<3>(v);
        return v[1];
    }
};

// exact median (rank 12) of 25 samples by forgetful selection: a working set of 14 loses its
// minimum and maximum each round, so after 11 insertions the 13th smallest is what is left
__device__ __forceinline__ int median25(const int* s)
{
    int v[14];
#pragma unroll
    for (int i = 0; i < 14; i++) v[i] = s[i];
    return Forget<14>::run(v, s + 14);
}

// generic rank selection for other window sizes (rank = nwin/2 like scipy's median_filter)
template <typename PX>
__device__ int median_generic(const PX* raw, int pitch, int r0, int c0, int med)
{
    int nwin = med * med, rank = nwin / 2;
    for (int a = 0; a < nwin; a++) {
        int va = raw[(r0 + a / med) * pitch + c0 + a % med];
        int less = 0, eq = 0;
        for (int b = 0; b < nwin; b++) {
            int vb = raw[(r0 + b / med) * pitch + c0 + b % med];
            less += (vb < va);
            eq += (vb == va);
        }
        if (less <= rank && rank < less + eq) return va;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K1. grid = (ceil(W/TW), ceil(H/TH), n_fields), block = 256.
// PX = the word the raw tile is staged in: unsigned short for the 16-bit formats, unsigned for FSQ_PIXELS_U32 (values < 2^31:
// pixel differences fit an int, products with a 32-bit matrix entry and their sum over the window fit an int64 by the
// domain check of fsq_detect).
template <bool FAST5, typename PX = unsigned short>
__global__ void __launch_bounds__(256) k1_response(const uint16_t* __restrict__ img, int pix_fmt, int H, int W, DetectConst dc,
                                                   long long* __restrict__ cm, unsigned long long* __restrict__ field_sum)
{
    extern __shared__ unsigned char smem[];
    const int med = dc.med, ksz = dc.ksz;
    const int mo = med / 2, me = med - 1 - mo;       // median window reaches [-mo, +me]
    const int kc = (ksz - 1) / 2;
    const int MH = TH + 2 * kc, MW = TW + 2 * kc;    // mf region
    const int RH = MH + mo + me, RW = MW + mo + me;  // raw region
    PX* raw = (PX*)smem;                                              // [RH][RW]
    int* mf = (int*)(smem + (((size_t)RH * RW * sizeof(PX) + 15) & ~(size_t)15));   // [MH][MW]
    const int f = blockIdx.z;
    const int h0 = blockIdx.y * TH, w0 = blockIdx.x * TW;
    const uint16_t* im = img + (size_t)f * H * W * (sizeof(PX) / 2);
    const int tid = threadIdx.x;

    for (int i = tid; i < RH * RW; i += 256) {       // raw tile with 'reflect' indexing
        int rr = i / RW, cc = i - rr * RW;
        int gh = reflect_idx(h0 - kc - mo + rr, H), gw = reflect_idx(w0 - kc - mo + cc, W);
        raw[i] = (PX)fsq_pixel(im, (size_t)gh * W + gw, pix_fmt);
    }
    __syncthreads();
    for (int i = tid; i < MH * MW; i += 256) {       // mf = img - min(median, img); 0 outside the image
        int rr = i / MW, cc = i - rr * MW;
        int gh = h0 - kc + rr, gw = w0 - kc + cc;
        int v = 0;
        if (gh >= 0 && gh < H && gw >= 0 && gw < W) {
            int m;
#ifdef FSQ_EXPERIMENT_NO_MEDIAN      // cost split of k1: the exact 25-sample median replaced by the centre's left neighbour (wrong results)
            if (FAST5) m = raw[(rr + mo) * RW + cc + mo - 1];
            else
#endif
            if (FAST5) {
                int s[25];
#pragma unroll
                for (int a = 0; a < 5; a++)
#pragma unroll
                    for (int b = 0; b < 5; b++) s[a * 5 + b] = raw[(rr + a) * RW + cc + b];
                m = median25(s);
            } else {
                m = median_generic(raw, RW, rr, cc, med);
            }
            int c = raw[(rr + mo) * RW + cc + mo];
            v = c - min(m, c);
        }
        mf[i] = v;
    }
    __syncthreads();
    unsigned long long local = 0;
    for (int i = tid; i < TH * TW; i += 256) {       // zero-padded correlation, clamp at 0
        int rr = i / TW, cc = i - rr * TW;
        int gh = h0 + rr, gw = w0 + cc;
        if (gh < H && gw < W) {
            long long s = 0;
            for (int a = 0; a < ksz; a++)
                for (int b = 0; b < ksz; b++) s += (long long)mf[(rr + a) * MW + cc + b] * (long long)dc.K[a * ksz + b];
            if (s < 0) s = 0;
            cm[((size_t)f * H + gh) * W + gw] = s;
            local += (unsigned long long)s;
        }
    }
    // block reduction of the exact integer sum, one atomic per block
    __shared__ unsigned long long red[4];
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((tid & 63) == 0) red[tid >> 6] = local;
    __syncthreads();
    if (tid == 0) atomicAdd(&field_sum[f], red[0] + red[1] + red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------
// K1 for the default 5 x 5 median window (round 4).  Round 3's kernel spent half its time in the 25-sample selection
// network (132 compare-exchanges per pixel on 25 LDS reads) and much of the rest in index arithmetic (a division and two
// reflections with a modulo per staged pixel).  Here:
//   * a block computes a fixed 16 x 64 region of the background-subtracted image mf (= 256 tasks of 4 horizontally adjacent
//     pixels, one per thread) and from it the (16 - 2 kc) x (64 - 2 kc) responses its correlation windows fit into;
//   * adjacent median windows share four of their five columns: a thread reads the 5 x 8 pixels of its four windows with ten
//     8-byte LDS reads, sorts the eight columns once (9 compare-exchanges each) and selects each median from five sorted
//     columns with the 66-comparator network of fsq_median_net.h (tools/gen_median_net.py: 108 min / max instructions);
//   * all tile geometry is compile-time constant; the 'reflect' index is two compares (the modulo form only for images
//     smaller than the halo);
//   * a correlation matrix whose entries depend on the ring only (the default one, pflib.py:48-52) is applied as
//     k_out (S5 - S3) + k_mid (S3 - c) + k_ctr c - the same integer, three multiplications instead of 25.
constexpr int M5R = 16, M5C = 64;                // mf region of a block
constexpr int R5R = M5R + 4, R5C = M5C + 4;      // raw region (median window reaches 2 pixels each way)

__device__ __forceinline__ int reflect_fast(int i, int n)
{
    int g = i < 0 ? -1 - i : i;
    g = g >= n ? 2 * n - 1 - g : g;
    return ((unsigned)g < (unsigned)n) ? g : reflect_idx(i, n);
}

template <bool RING5>
__global__ void __launch_bounds__(256) k1_response5(const uint16_t* __restrict__ img, int pix_fmt, int H, int W, DetectConst dc,
                                                    long long* __restrict__ cm, unsigned long long* __restrict__ field_sum)
{
    __shared__ __attribute__((aligned(16))) unsigned short raw[R5R * R5C];
    __shared__ int mf[M5R * M5C];
    const int ksz = RING5 ? 5 : dc.ksz, kc = (ksz - 1) / 2;
    const int TH5 = M5R - 2 * kc, TW5 = M5C - 2 * kc;              // responses per block
    const int f = blockIdx.z, tid = threadIdx.x;
    const int h0 = blockIdx.y * TH5, w0 = blockIdx.x * TW5;
    const uint16_t* im = img + (size_t)f * H * W;
    for (int i = tid; i < R5R * R5C; i += 256) {                    // raw tile, scipy's 'reflect' indexing
        const int rr = i / R5C, cc = i - rr * R5C;
        const int gh = reflect_fast(h0 - kc - 2 + rr, H), gw = reflect_fast(w0 - kc - 2 + cc, W);
        raw[i] = (unsigned short)fsq_pixel(im, (size_t)gh * W + gw, pix_fmt);
    }
    __syncthreads();
    {   // mf = img - min(median, img); 0 outside the image (the correlation is zero padded)
        const int r = tid >> 4, g4 = (tid & 15) * 4;
        int col[8][5], ctr[4];
#pragma unroll
        for (int a = 0; a < 5; a++) {
            const uint2* p = (const uint2*)(raw + (r + a) * R5C + g4);     // 8-byte aligned: R5C * 2 and g4 * 2 are multiples of 8
            const uint2 lo = p[0], hi = p[1];
            const unsigned w8[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int b = 0; b < 8; b++) col[b][a] = (int)((w8[b >> 1] >> (16 * (b & 1))) & 0xffffu);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) ctr[k] = col[k + 2][2];
#pragma unroll
        for (int b = 0; b < 8; b++) fsq_sort5(col[b][0], col[b][1], col[b][2], col[b][3], col[b][4]);
        const int gh = h0 - kc + r;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int v[25];
#pragma unroll
            for (int b = 0; b < 5; b++)
#pragma unroll
                for (int a = 0; a < 5; a++) v[b * 5 + a] = col[k + b][a];
            const int m = fsq_median_of_sorted_columns(v);
            const int gw = w0 - kc + g4 + k;
            const bool inside = gh >= 0 && gh < H && gw >= 0 && gw < W;
            mf[r * M5C + g4 + k] = inside ? ctr[k] - min(m, ctr[k]) : 0;
        }
    }
    __syncthreads();
    unsigned long long local = 0;
    for (int i = tid; i < TH5 * TW5; i += 256) {                     // zero-padded correlation, clamp at 0
        const int rr = i / TW5, cc = i - rr * TW5;
        const int gh = h0 + rr, gw = w0 + cc;
        if (gh < H && gw < W) {
            long long s;
            if (RING5) {
                int s5 = 0, s3 = 0;
#pragma unroll
                for (int a = 0; a < 5; a++)
#pragma unroll
                    for (int b = 0; b < 5; b++) {
                        const int v = mf[(rr + a) * M5C + cc + b];
                        s5 += v;
                        if (a >= 1 && a <= 3 && b >= 1 && b <= 3) s3 += v;
                    }
                const int c0 = mf[(rr + 2) * M5C + cc + 2];
                s = (long long)dc.K[0] * (long long)(s5 - s3) + (long long)dc.K[6] * (long long)(s3 - c0) + (long long)dc.K[12] * (long long)c0;
            } else {
                s = 0;
                for (int a = 0; a < ksz; a++)
                    for (int b = 0; b < ksz; b++) s += (long long)mf[(rr + a) * M5C + cc + b] * (long long)dc.K[a * ksz + b];
            }
            if (s < 0) s = 0;
            cm[((size_t)f * H + gh) * W + gw] = s;
            local += (unsigned long long)s;
        }
    }
    __shared__ unsigned long long red[4];
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((tid & 63) == 0) red[tid >> 6] = local;
    __syncthreads();
    if (tid == 0) atomicAdd(&field_sum[f], red[0] + red[1] + red[2] + red[3]);
}

// numpy pairwise sum (loops_utils.h.src) of n <= 8192 squared deviations, serial (ragged last chunk)
__device__ double sqdev(const long long* cm, double mean, int i)
{
    double d = (double)cm[i] - mean;
    return d * d;
}
__device__ double pairwise_leaf(const long long* cm, double mean, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += sqdev(cm, mean, i);
        return res;
    }
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = sqdev(cm, mean, k);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += sqdev(cm, mean, i + k);
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += sqdev(cm, mean, i);
    return res;
}
__device__ double pairwise_serial(const long long* cm, double mean, int n)
{   // explicit stack instead of recursion; depth <= log2(8192/128)+1
    int st_off[16], st_n[16], st_state[16];
    double st_left[16];
    int sp = 0;
    st_off[0] = 0; st_n[0] = n; st_state[0] = 0;
    double ret = 0.;
    while (sp >= 0) {
        int off = st_off[sp], nn = st_n[sp];
        if (nn <= 128) { ret = pairwise_leaf(cm + off, mean, nn); sp--; continue; }
        int n2 = nn / 2;
        n2 -= n2 % 8;
        if (st_state[sp] == 0) { st_state[sp] = 1; sp++; st_off[sp] = off; st_n[sp] = n2; st_state[sp] = 0; }
        else if (st_state[sp] == 1) { st_left[sp] = ret; st_state[sp] = 2; sp++; st_off[sp] = off + n2; st_n[sp] = nn - n2; st_state[sp] = 0; }
        else { ret = st_left[sp] + ret; sp--; }
    }
    return ret;
}

// K2a. grid = (n_chunks, n_fields), block = 256.  chunk_sum[f][c] = pairwise sum over the chunk.
__global__ void __launch_bounds__(256) k2_sqdev_chunks(const long long* __restrict__ cm, long long N,
                                                       const unsigned long long* __restrict__ field_sum,
                                                       int n_chunks, double* __restrict__ chunk_sum)
{
    const int f = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const double mean = (double)field_sum[f] / (double)N;          // numpy.mean: exact integer sum / N
    const long long base = (long long)c * CHUNK;
    const long long* a = cm + (size_t)f * N + base;
    long long m = N - base;
    if (m > CHUNK) m = CHUNK;
    __shared__ double r[64][8];
    if (m == CHUNK) {
        // 64 leaves of 128; thread t owns accumulators (2s, 2s+1) of leaf t/4: r[k] = a[k] + a[8+k] + ...
        const int leaf = tid >> 2, s = tid & 3;
        const long long* p = a + leaf * 128 + 2 * s;
        double r0, r1;
        { double d0 = (double)p[0] - mean, d1 = (double)p[1] - mean; r0 = d0 * d0; r1 = d1 * d1; }
#pragma unroll 5
        for (int i = 1; i < 16; i++) {
            double d0 = (double)p[8 * i] - mean, d1 = (double)p[8 * i + 1] - mean;
            r0 += d0 * d0;
            r1 += d1 * d1;
        }
        r[leaf][2 * s] = r0;
        r[leaf][2 * s + 1] = r1;
        __syncthreads();
        __shared__ double tree[64];
        if (tid < 64) tree[tid] = ((r[tid][0] + r[tid][1]) + (r[tid][2] + r[tid][3])) + ((r[tid][4] + r[tid][5]) + (r[tid][6] + r[tid][7]));
        __syncthreads();
        for (int w = 1; w < 64; w <<= 1) {            // balanced tree: left + right
            double v = 0.;
            bool act = (tid < 64) && ((tid & (2 * w - 1)) == 0);
            if (act) v = tree[tid] + tree[tid + w];
            __syncthreads();
            if (act) tree[tid] = v;
            __syncthreads();
        }
        if (tid == 0) chunk_sum[(size_t)f * n_chunks + c] = tree[0];
    } else if (tid == 0) {
        chunk_sum[(size_t)f * n_chunks + c] = pairwise_serial(a, mean, (int)m);
    }
}

// K2b. one thread per field.
__global__ void k2_threshold(const unsigned long long* __restrict__ field_sum, const double* __restrict__ chunk_sum,
                             int n_chunks, long long N, double c_std, int n_fields, double* __restrict__ thr)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_fields) return;
    double mean = (double)field_sum[f] / (double)N;
    double s = 0.0;
    for (int c = 0; c < n_chunks; c++) s += chunk_sum[(size_t)f * n_chunks + c];
    double var = s / (double)N;
    thr[f] = mean + c_std * __builtin_sqrt(var);      // pflib.py:250
}

__device__ __forceinline__ bool is_candidate(const long long* cmf, long long i, int H, int W, double thr, int* ph, int* pw)
{
    int h = (int)(i / W), w = (int)(i - (long long)h * W);
    *ph = h; *pw = w;
    if (h < 2 || h >= H - 2 || w < 2 || w >= W - 2) return false;     // pflib.py:252-253
    return !((double)cmf[i] < thr);                                   // pflib.py:254
}

// K2c pass 1. grid = (tiles, n_fields), block 256, 1024 px per tile.
__global__ void __launch_bounds__(256) k2_count(const long long* __restrict__ cm, int H, int W, const double* __restrict__ thr,
                                                int tiles, int* __restrict__ tile_count)
{
    const int f = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const long long N = (long long)H * W;
    const long long* cmf = cm + (size_t)f * N;
    const double th = thr[f];
    int cnt = 0;
    for (int k = 0; k < 4; k++) {
        long long i = (long long)t * 1024 + k * 256 + tid;
        int h, w;
        if (i < N && is_candidate(cmf, i, H, W, th, &h, &w)) cnt++;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    __shared__ int red[4];
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) tile_count[(size_t)f * tiles + t] = red[0] + red[1] + red[2] + red[3];
}

// pass 2. one block per field: exclusive scan of its tile counts (in place) + field total
__global__ void __launch_bounds__(256) k2_scan_field(int* __restrict__ tile_count, int tiles, int* __restrict__ counts)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    int* tc = tile_count + (size_t)f * tiles;
    __shared__ int part[256];
    __shared__ int carry;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < tiles; base += 256) {
        int i = base + tid;
        int v = (i < tiles) ? tc[i] : 0;
        part[tid] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {           // Hillis-Steele inclusive scan
            int add = (tid >= o) ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        int excl = part[tid] - v + carry;
        if (i < tiles) tc[i] = excl;
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) counts[f] = carry;
}

// pass 3. single block: exclusive scan of the field totals -> offsets[0..n_fields], counts[n_fields] = total
// Also the exactness check of the threshold: numpy.mean sums the int64 response in float64, which equals the exact
// integer sum used here as long as that sum stays below 2^53 (all terms are >= 0, so every partial sum does too).
// A field beyond that makes the total -1, which the caller reports as "not implemented" instead of a wrong answer.
__global__ void __launch_bounds__(256) k2_scan_all(int* __restrict__ counts, int* __restrict__ offsets, int n_fields,
                                                   const unsigned long long* __restrict__ field_sum)
{
    const int tid = threadIdx.x;
    __shared__ int part[256];
    __shared__ int carry, inexact;
    if (tid == 0) { carry = 0; inexact = 0; }
    __syncthreads();
    for (int i = tid; i < n_fields; i += 256)
        if (field_sum[i] >= (1ull << 53)) inexact = 1;
    for (int base = 0; base < n_fields; base += 256) {
        int i = base + tid;
        int v = (i < n_fields) ? counts[i] : 0;
        part[tid] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            int add = (tid >= o) ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (i < n_fields) offsets[i] = part[tid] - v + carry;
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) { offsets[n_fields] = carry; counts[n_fields] = inexact ? -1 : carry; }
}

// pass 4. same grid as pass 1: ordered write of (field, h, w)
__global__ void __launch_bounds__(256) k2_write(const long long* __restrict__ cm, int H, int W, const double* __restrict__ thr,
                                                int tiles, const int* __restrict__ tile_off, const int* __restrict__ offsets,
                                                int* __restrict__ cand, long long cap)
{
    const int f = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const long long N = (long long)H * W;
    const long long* cmf = cm + (size_t)f * N;
    const double th = thr[f];
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int wave_cnt[4];
    long long out = (long long)offsets[f] + tile_off[(size_t)f * tiles + t];
    for (int k = 0; k < 4; k++) {                     // 256 consecutive pixels per round keep raster order
        long long i = (long long)t * 1024 + k * 256 + tid;
        int h = 0, w = 0;
        bool flag = (i < N) && is_candidate(cmf, i, H, W, th, &h, &w);
        unsigned long long bal = __ballot(flag);
        int rank = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
        for (int q = 0; q < 4; q++) { int c = wave_cnt[q]; if (q < wave) before += c; total += c; }
        long long pos = out + before + rank;
        if (flag && pos < cap) { cand[3 * pos] = f; cand[3 * pos + 1] = h; cand[3 * pos + 2] = w; }
        out += total;
        __syncthreads();
    }
}

struct WsLayout {
    size_t cm, field_sum, chunk_sum, thr, tile_count, total;
};
WsLayout ws_layout(int n_fields, int H, int W)
{
    WsLayout L;
    size_t N = (size_t)H * W;
    size_t n_chunks = (N + CHUNK - 1) / CHUNK, tiles = (N + 1023) / 1024;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.cm = take((size_t)n_fields * N * 8);
    L.field_sum = take((size_t)n_fields * 8);
    L.chunk_sum = take((size_t)n_fields * n_chunks * 8);
    L.thr = take((size_t)n_fields * 8);
    L.tile_count = take((size_t)n_fields * tiles * 4);
    L.total = o;
    return L;
}

}  // namespace

extern "C" int64_t fsq_detect_workspace_bytes(int n_fields, int H, int W)
{
    if (n_fields < 1 || H < 1 || W < 1) return FSQ_EINVAL;
    return (int64_t)ws_layout(n_fields, H, W).total;
}

extern "C" int fsq_detect(const uint16_t* d_img, int n_fields, int H, int W, const FsqDetectParams* prm,
                          int32_t* d_cand, int64_t cap, int32_t* d_counts, int32_t* d_offsets, double* d_thr,
                          void* d_workspace, int64_t workspace_bytes, void* stream)
{
    if (!prm) return FSQ_EINVAL;
    const int med = prm->median_filter_size, ksz = prm->ksz;
    if (ksz < 1 || (ksz % 2) == 0) return FSQ_EINVAL;                 // pflib.py:236-239 -> ValueError
    if (prm->pixel_format != FSQ_PIXELS_U16 && prm->pixel_format != FSQ_PIXELS_F16 && prm->pixel_format != FSQ_PIXELS_U32) return FSQ_EINVAL;
    const bool wide = prm->pixel_format == FSQ_PIXELS_U32;
    if (wide && (prm->pixel_bits < 0 || prm->pixel_bits > 31)) return FSQ_EINVAL;
    if (ksz > MAXK || med < 1 || med > MAXK) return FSQ_ENOTIMPL;
    if (n_fields < 1 || H < 1 || W < 1 || cap < 0 || !d_img || !d_counts || !d_offsets || !d_workspace) return FSQ_EINVAL;
    if (cap > 0 && !d_cand) return FSQ_EINVAL;
    if ((long long)H * W > (1ll << 31)) return FSQ_ENOTIMPL;
    WsLayout L = ws_layout(n_fields, H, W);
    if ((size_t)workspace_bytes < L.total) return FSQ_ENOMEM;
    DetectConst dc;
    dc.med = med; dc.ksz = ksz;
    for (int i = 0; i < ksz * ksz; i++) {
        if (prm->K[i] > 2147483647ll || prm->K[i] < -2147483648ll) return FSQ_ENOTIMPL;
        dc.K[i] = (int)prm->K[i];
    }
    // exactness domain: the integer sum of the response must not wrap (its float64 exactness, < 2^53, is checked
    // on the actual data by k2_scan_all)
    {
        long double kmax = 0, kabs = 0;
        for (int i = 0; i < ksz * ksz; i++) {
            if (dc.K[i] > 0) kmax += dc.K[i];
            kabs += dc.K[i] > 0 ? (long double)dc.K[i] : -(long double)dc.K[i];
        }
        const int bits = wide ? (prm->pixel_bits ? prm->pixel_bits : 31) : 16;
        const long double pmax = (long double)((1ull << bits) - 1);
        if (kabs * pmax >= 9223372036854775807.0L) return FSQ_ENOTIMPL;                 // one pixel's window sum (int64)
        if (kmax * pmax * (long double)H * W >= 18446744073709551615.0L) return FSQ_ENOTIMPL;
    }
    hipStream_t s = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)d_workspace;
    long long* cm = (long long*)(ws + L.cm);
    unsigned long long* field_sum = (unsigned long long*)(ws + L.field_sum);
    double* chunk_sum = (double*)(ws + L.chunk_sum);
    double* thr = d_thr ? d_thr : (double*)(ws + L.thr);
    int* tile_count = (int*)(ws + L.tile_count);
    const long long N = (long long)H * W;
    const int n_chunks = (int)((N + CHUNK - 1) / CHUNK), tiles = (int)((N + 1023) / 1024);

    FSQ_HIP_CHECK(hipMemsetAsync(field_sum, 0, (size_t)n_fields * 8, s));
    {
        const int mo = med / 2, me = med - 1 - mo, kc = (ksz - 1) / 2;
        const int MH = TH + 2 * kc, MW = TW + 2 * kc, RH = MH + mo + me, RW = MW + mo + me;
        size_t shm = (((size_t)RH * RW * (wide ? 4 : 2) + 15) & ~(size_t)15) + (size_t)MH * MW * 4;
        dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, n_fields);
        if (wide) {           // 32-bit pixels: the generic tile kernel with a 32-bit raw tile
            if (med == 5) hipLaunchKernelGGL((k1_response<true, unsigned>), grid, dim3(256), shm, s, d_img, prm->pixel_format, H, W, dc, cm, field_sum);
            else hipLaunchKernelGGL((k1_response<false, unsigned>), grid, dim3(256), shm, s, d_img, prm->pixel_format, H, W, dc, cm, field_sum);
        } else if (med == 5 && ksz <= 9 && !getenv("FSQ_DETECT_R03")) {
            // a correlation matrix whose entries depend on the ring only (the default one) needs three multiplications per pixel
            bool ring = (ksz == 5);
            for (int a = 0; a < 5 && ring; a++)
                for (int b = 0; b < 5; b++) {
                    const int d = std::max(abs(a - 2), abs(b - 2));
                    if (dc.K[a * 5 + b] != (d == 2 ? dc.K[0] : d == 1 ? dc.K[6] : dc.K[12])) ring = false;
                }
            const int th5 = M5R - 2 * kc, tw5 = M5C - 2 * kc;
            dim3 grid5((W + tw5 - 1) / tw5, (H + th5 - 1) / th5, n_fields);
            if (ring) hipLaunchKernelGGL(k1_response5<true>, grid5, dim3(256), 0, s, d_img, prm->pixel_format, H, W, dc, cm, field_sum);
            else hipLaunchKernelGGL(k1_response5<false>, grid5, dim3(256), 0, s, d_img, prm->pixel_format, H, W, dc, cm, field_sum);
        } else if (med == 5) hipLaunchKernelGGL(k1_response<true>, grid, dim3(256), shm, s, d_img, prm->pixel_format, H, W, dc, cm, field_sum);
        else hipLaunchKernelGGL(k1_response<false>, grid, dim3(256), shm, s, d_img, prm->pixel_format, H, W, dc, cm, field_sum);
    }
    hipLaunchKernelGGL(k2_sqdev_chunks, dim3(n_chunks, n_fields), dim3(256), 0, s, cm, N, field_sum, n_chunks, chunk_sum);
    hipLaunchKernelGGL(k2_threshold, dim3((n_fields + 63) / 64), dim3(64), 0, s, field_sum, chunk_sum, n_chunks, N,
                       prm->c_std, n_fields, thr);
    hipLaunchKernelGGL(k2_count, dim3(tiles, n_fields), dim3(256), 0, s, cm, H, W, thr, tiles, tile_count);
    hipLaunchKernelGGL(k2_scan_field, dim3(n_fields), dim3(256), 0, s, tile_count, tiles, d_counts);
    hipLaunchKernelGGL(k2_scan_all, dim3(1), dim3(256), 0, s, d_counts, d_offsets, n_fields, field_sum);
    hipLaunchKernelGGL(k2_write, dim3(tiles, n_fields), dim3(256), 0, s, cm, H, W, thr, tiles, tile_count, d_offsets,
                       d_cand, (long long)cap);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}
