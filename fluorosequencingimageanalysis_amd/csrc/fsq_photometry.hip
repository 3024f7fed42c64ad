// fsq_photometry.hip - spot photometry on the peak table, gfx950.
//
// fsq_mexican_hat: Spot.mexican_hat_photometry_metric (flexlibrary.py:172-210) for a table of integer spot centres.
// One wavefront per spot: the (2*radius+1)^2 window, clipped at the image borders exactly like Spot.image_slice
// (flexlibrary.py:140-146), is spread over the lanes (<= 16 pixels per lane, held in registers); crown pixels are
// summed (exact integer), the median of the brim is found by a 16-step binary search on the pixel VALUE with a
// wave-wide count per step (no sort, no LDS) - twice for an even count (numpy.median = mean of the two middle values).
// HBM-bound by construction: (2r+1)^2 x 2 B read per spot, 8 B written.
#include "fsq_common.h"

namespace {

constexpr int MAXR = 15;                    // 31 x 31 = 961 pixels <= 16 per lane
constexpr int PER_LANE = 16;

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// smallest value x such that #{brim <= x} >= rank + 1   (rank is 0-based); BITS = 16 for uint16 pixels, 31 for uint32 (< 2^31)
template <int BITS>
__device__ __forceinline__ unsigned kth_smallest(const unsigned (&v)[PER_LANE], int rank)
{
    unsigned lo = 0, hi = (1u << BITS) - 1u;
    for (int it = 0; it < BITS; it++) {
        const unsigned mid = (lo + hi) >> 1;
        int c = 0;
#pragma unroll
        for (int t = 0; t < PER_LANE; t++) c += (v[t] <= mid);
        c = wave_sum_i(c);
        if (c >= rank + 1) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// PX: uint16_t, or uint32_t for FSQ_PIXELS_U32 frames (values < 2^31: the search runs over 31 bits)
template <typename PX>
__global__ void __launch_bounds__(256) k7_mexican_hat(const PX* __restrict__ img, int H, int W,
                                                       const int32_t* __restrict__ fhw, long long n, int brim, int radius,
                                                       double* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long spot = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (spot >= n) return;
    const int f = fhw[3 * spot], h = fhw[3 * spot + 1], w = fhw[3 * spot + 2];
    const int r0 = max(0, h - radius), r1 = min(H, h + radius + 1);
    const int c0 = max(0, w - radius), c1 = min(W, w + radius + 1);
    const int hc = max(r1 - r0, 0), wc = max(c1 - c0, 0), npx = hc * wc, diameter = 2 * radius + 1;
    constexpr int BITS = sizeof(PX) == 2 ? 16 : 31;
    const PX* base = img + ((size_t)f * H) * W;
    unsigned v[PER_LANE];
    long long crown = 0;
    int ncrown = 0, nbrim = 0;
#pragma unroll
    for (int t = 0; t < PER_LANE; t++) {
        const int i = lane + 64 * t;
        v[t] = 1u << BITS;                                  // not a brim pixel: above every pixel value
        if (i < npx) {
            const int hh = i / wc, ww = i - hh * wc;
            const unsigned p = base[(size_t)(r0 + hh) * W + (c0 + ww)];
            const bool in_crown = (brim <= hh) && (hh < diameter - brim) && (brim <= ww) && (ww < diameter - brim);
            if (in_crown) { crown += p; ncrown++; }
            else { v[t] = p; nbrim++; }
        }
    }
    crown = wave_sum_ll(crown);
    ncrown = wave_sum_i(ncrown);
    nbrim = wave_sum_i(nbrim);
    double med;
    if (nbrim == 0) med = __builtin_nan("");
    else if (nbrim & 1) med = (double)kth_smallest<BITS>(v, nbrim / 2);
    else med = ((double)kth_smallest<BITS>(v, nbrim / 2 - 1) + (double)kth_smallest<BITS>(v, nbrim / 2)) / 2.0;
    if (lane == 0) out[spot] = (double)crown - (double)ncrown * med;
}

// Any radius (round 4: the register form above holds 31 x 31 windows; the reference has no limit, flexlibrary.py:172-210): the
// same arithmetic with the window re-read from memory (it stays in L1 / L2) for every step of the binary search.
template <typename PX>
__global__ void __launch_bounds__(256) k7_mexican_hat_any(const PX* __restrict__ img, int H, int W,
                                                           const int32_t* __restrict__ fhw, long long n, int brim, int radius,
                                                           double* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long spot = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (spot >= n) return;
    const int f = fhw[3 * spot], h = fhw[3 * spot + 1], w = fhw[3 * spot + 2];
    const int r0 = max(0, h - radius), r1 = min(H, h + radius + 1);
    const int c0 = max(0, w - radius), c1 = min(W, w + radius + 1);
    const int hc = max(r1 - r0, 0), wc = max(c1 - c0, 0), npx = hc * wc, diameter = 2 * radius + 1;
    constexpr int BITS = sizeof(PX) == 2 ? 16 : 31;
    const PX* base = img + ((size_t)f * H) * W;
    auto in_crown = [&](int hh, int ww) { return (brim <= hh) && (hh < diameter - brim) && (brim <= ww) && (ww < diameter - brim); };
    long long crown = 0;
    int ncrown = 0, nbrim = 0;
    for (int i = lane; i < npx; i += 64) {
        const int hh = i / wc, ww = i - hh * wc;
        const unsigned p = base[(size_t)(r0 + hh) * W + (c0 + ww)];
        if (in_crown(hh, ww)) { crown += p; ncrown++; } else nbrim++;
    }
    crown = wave_sum_ll(crown);
    ncrown = wave_sum_i(ncrown);
    nbrim = wave_sum_i(nbrim);
    auto kth = [&](int rank) {           // smallest value x with #{brim <= x} >= rank + 1
        unsigned lo = 0, hi = (1u << BITS) - 1u;
        for (int it = 0; it < BITS; it++) {
            const unsigned mid = (lo + hi) >> 1;
            int c = 0;
            for (int i = lane; i < npx; i += 64) {
                const int hh = i / wc, ww = i - hh * wc;
                if (!in_crown(hh, ww)) c += ((unsigned)base[(size_t)(r0 + hh) * W + (c0 + ww)] <= mid);
            }
            c = wave_sum_i(c);
            if (c >= rank + 1) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    double med;
    if (nbrim == 0) med = __builtin_nan("");
    else if (nbrim & 1) med = (double)kth(nbrim / 2);
    else med = ((double)kth(nbrim / 2 - 1) + (double)kth(nbrim / 2)) / 2.0;
    if (lane == 0) out[spot] = (double)crown - (double)ncrown * med;
}

}  // namespace

template <typename PX>
static int mexican_hat_launch(const PX* d_img, int n_fields, int H, int W, const int32_t* d_fhw, int64_t n, int brim_size, int radius,
                              double* d_out, void* stream)
{
    if (n < 0 || n_fields < 1 || H < 1 || W < 1 || brim_size < 0 || radius < 0) return FSQ_EINVAL;
    if (radius > 16383) return FSQ_EINVAL;                        // ((2 r + 1)^2 must fit an int)
    if (n == 0) return FSQ_OK;
    if (!d_img || !d_fhw || !d_out) return FSQ_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (radius <= MAXR)
        hipLaunchKernelGGL(k7_mexican_hat<PX>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, d_img, H, W, d_fhw, (long long)n,
                           brim_size, radius, d_out);
    else
        hipLaunchKernelGGL(k7_mexican_hat_any<PX>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, d_img, H, W, d_fhw, (long long)n,
                           brim_size, radius, d_out);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}

extern "C" int fsq_mexican_hat(const uint16_t* d_img, int n_fields, int H, int W, const int32_t* d_fhw, int64_t n,
                               int brim_size, int radius, double* d_out, void* stream)
{
    return mexican_hat_launch(d_img, n_fields, H, W, d_fhw, n, brim_size, radius, d_out, stream);
}

extern "C" int fsq_mexican_hat_u32(const uint32_t* d_img, int n_fields, int H, int W, const int32_t* d_fhw, int64_t n,
                                   int brim_size, int radius, double* d_out, void* stream)
{
    return mexican_hat_launch(d_img, n_fields, H, W, d_fhw, n, brim_size, radius, d_out, stream);
}
