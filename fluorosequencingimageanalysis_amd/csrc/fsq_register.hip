// fsq_register.hip - K6: sub-pixel frame registration by phase correlation, batched over image pairs.
// Reference: phase_correlate.phase_correlate / _dftups, phase_correlate.py:11-196 (Guizar-Sicairos et al.).
//
//   F = fft2(ref), G = fft2(reg)                       rocFFT (hipFFT front end), fp64, batched.  The images are real, so
//                                                       the transforms are real-to-complex: HALF spectra H x (W/2+1)
//   cc = ifft2(F * conj(G))                             cross-power on the half spectrum + complex-to-real inverse (the
//                                                       cross-correlation of real images is real)
//   (row_max, col_max) = argmax(cc)                     first maximum like numpy
//   upsample_factor > 1: matrix-multiply DFT of G*conj(F) on a ceil(1.5*uf)^2 grid around the peak,
//                        argmax again, error / diffphase from the peak value       (:94-122)
// Everything after the inputs stays on the device: peak -> shift -> DFT offsets -> error / diffphase are small kernels,
// the call only enqueues.  The kernels are HBM-streaming (conversion, cross-power, reductions) except the upsampled DFT,
// whose row product is a complex GEMM on v_mfma_f64_16x16x4_f64 (k6_dft_mfma below; its B operand G*conj(F) is formed on
// the fly from the two half spectra, mirrored columns by Hermitian symmetry); everything in fp64 like the reference.
// FSQ_REGISTER_Z2Z=1 selects the older full-spectrum complex-to-complex path (A/B; also the fallback for shapes rocFFT
// has no real-transform plan for).
#include <hipfft/hipfft.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "fsq_common.h"

namespace {

typedef double2 cplx;

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }

__global__ void k6_to_complex(const double* __restrict__ a, cplx* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_double2(a[i], 0.0);
}
__global__ void k6_u16_to_complex(const uint16_t* __restrict__ a, cplx* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_double2((double)a[i], 0.0);
}

// prod = F * conj(G)   (phase_correlate.py:71); also per-pair sums of |F|^2, |G|^2 (:86-87, 117-120)
__global__ void __launch_bounds__(256) k6_cross_power(const cplx* __restrict__ F, const cplx* __restrict__ G, cplx* __restrict__ prod,
                                                      size_t npix, double* __restrict__ sums /*[pairs][2]*/)
{
    const int pair = blockIdx.y;
    const cplx* f = F + (size_t)pair * npix;
    const cplx* g = G + (size_t)pair * npix;
    cplx* p = prod + (size_t)pair * npix;
    double sf = 0., sg = 0.;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        cplx a = f[i], b = g[i];
        p[i] = cmul(a, cconj(b));
        sf += a.x * a.x + a.y * a.y;
        sg += b.x * b.x + b.y * b.y;
    }
    __shared__ double r0[256], r1[256];
    r0[threadIdx.x] = sf; r1[threadIdx.x] = sg;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    // (partial sums per block, added up in block order by k6_coarse: the same bits on every run)
    if (threadIdx.x == 0) { sums[((size_t)pair * gridDim.x + blockIdx.x) * 2] = r0[0]; sums[((size_t)pair * gridDim.x + blockIdx.x) * 2 + 1] = r1[0]; }
}

struct Peak { double re, im; long long idx; };
__device__ __forceinline__ bool peak_better(const Peak& a, const Peak& b)
{   // numpy argmax on complex: lexicographic (real, imag); first occurrence wins ties
    if (a.re != b.re) return a.re > b.re;
    if (a.im != b.im) return a.im > b.im;
    return a.idx < b.idx;
}

// one block per pair: argmax of scale * data (scale real > 0 keeps the order); optional conjugation
__global__ void __launch_bounds__(256) k6_argmax(const cplx* __restrict__ data, size_t n, double scale, int conj, Peak* __restrict__ out)
{
    const int pair = blockIdx.x;
    const cplx* d = data + (size_t)pair * n;
    Peak best;
    best.re = -__builtin_inf(); best.im = -__builtin_inf(); best.idx = (long long)n;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        Peak p;
        p.re = d[i].x * scale; p.im = (conj ? -d[i].y : d[i].y) * scale; p.idx = (long long)i;
        if (p.re != p.re) continue;                    // NaN never wins (numpy would propagate; inputs are finite)
        if (peak_better(p, best)) best = p;
    }
    __shared__ Peak sh[256];
    sh[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o && peak_better(sh[threadIdx.x + o], sh[threadIdx.x])) sh[threadIdx.x] = sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[pair] = sh[0];
}

// prod2 = G * conj(F) (phase_correlate.py:102)
__global__ void k6_cross_power2(const cplx* __restrict__ F, const cplx* __restrict__ G, cplx* __restrict__ prod, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) prod[i] = cmul(G[i], cconj(F[i]));
}

__device__ __forceinline__ double fftfreq_idx(int i, int n) { return (double)((i + n / 2) % n) - floor(n / 2.0); }  // ifftshift(arange(n)) - floor(n/2)

// _dftups (phase_correlate.py:137-196): out[u][v] = sum_r sum_c rk[u][r] * data[r][c] * ck[c][v].
// grid = (up, pairs); block = 256; dynamic LDS = (DFT_RC + DFT_TC) complex - ANY rows x cols.
//   stage 1: the row kernel rk[u][.] of this block's u is evaluated once (rows sincos instead of rows x cols), DFT_RC rows
//            at a time into LDS; T[c] = sum_r rk[u][r] * data[r][c], thread per column, r in order over all row chunks
//            (coalesced reads of data).  T lives in LDS when cols <= DFT_TC, else in the caller's scratch Tg - every
//            thread only ever touches its own columns, so the partial sums carry over between chunks without hazards;
//   stage 2: out[u][v] = sum_c T[c] * ck[c][v]: every thread takes the columns c = tid, tid + 256, ..., the partial sums
//            are combined by a fixed-shape tree (wave shuffles, then the 4 waves in order) - deterministic.
constexpr int DFT_RC = 1536, DFT_TC = 1536;
__global__ void __launch_bounds__(256) k6_dftups(const cplx* __restrict__ data, int rows, int cols, int up, int uf,
                                                 const double* __restrict__ offs /*[pairs][2] row_off, col_off*/,
                                                 cplx* __restrict__ out /*[pairs][up][up]*/, cplx* __restrict__ Tg /*[pairs][up][cols] or null*/)
{
    extern __shared__ cplx lds_c[];
    cplx* Wr = lds_c;                                 // [min(rows, DFT_RC)]
    __shared__ cplx wsum[4];
    const int u = blockIdx.x, pair = blockIdx.y, tid = threadIdx.x;
    cplx* T = (cols <= DFT_TC) ? lds_c + DFT_RC : Tg + ((size_t)pair * up + u) * cols;
    const cplx* d = data + (size_t)pair * rows * cols;
    const double roff = offs[2 * pair], coff = offs[2 * pair + 1];
    const double cr = -2.0 * 3.141592653589793 / ((double)rows * uf), cc = -2.0 * 3.141592653589793 / ((double)cols * uf);
    for (int c = tid; c < cols; c += blockDim.x) T[c] = make_double2(0., 0.);
    for (int r0 = 0; r0 < rows; r0 += DFT_RC) {
        const int nr = min(DFT_RC, rows - r0);
        __syncthreads();
        for (int r = tid; r < nr; r += blockDim.x) {
            double ph = cr * (((double)u - roff) * fftfreq_idx(r0 + r, rows));
            double s, co;
            sincos(ph, &s, &co);
            Wr[r] = make_double2(co, s);
        }
        __syncthreads();
        for (int c = tid; c < cols; c += blockDim.x) {
            cplx acc = T[c];
            for (int r = 0; r < nr; r++) {
                const cplx w = Wr[r];
                const cplx v = d[(size_t)(r0 + r) * cols + c];
                acc.x += w.x * v.x - w.y * v.y;
                acc.y += w.x * v.y + w.y * v.x;
            }
            T[c] = acc;
        }
    }
    __syncthreads();
    for (int v = 0; v < up; v++) {
        cplx acc = make_double2(0., 0.);
        for (int c = tid; c < cols; c += blockDim.x) {
            double ph = cc * (fftfreq_idx(c, cols) * ((double)v - coff));
            double s, co;
            sincos(ph, &s, &co);
            const cplx t = T[c];
            acc.x += t.x * co - t.y * s;
            acc.y += t.x * s + t.y * co;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); }
        if ((tid & 63) == 0) wsum[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            cplx t = wsum[0];
            for (int k = 1; k < 4; k++) { t.x += wsum[k].x; t.y += wsum[k].y; }
            out[((size_t)pair * up + u) * up + v] = t;
        }
        __syncthreads();
    }
}

// ---- the same product on the matrix cores -------------------------------------------------------------------------
// T[u][c] = sum_r rk[u][r] * data[r][c] is a (up x rows) x (rows x cols) complex GEMM - the one dense contraction of
// the whole path - so it runs on v_mfma_f64_16x16x4_f64 (fp64 in, fp64 accumulate: same precision as the reference's
// numpy.dot): k6_rowkernel writes rk transposed and zero-padded to a multiple of 16 rows, rkT[r][u]; one wave of
// k6_dft_mfma owns a 32 x 16 block of T (2 M-tiles, real and imaginary accumulators = 16 doubles per lane) and walks
// r in steps of 4, each step being 8 MFMAs fed by three 16-byte loads per lane (A: rkT[r0 + l/16][16 mt + l%16],
// B: data[r0 + l/16][c0 + l%16]; C/D: column l%16, row l/16 + 4 reg).  k6_dft_cols then applies the column kernel.
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k6_rowkernel(int rows, int up, int Mp, int uf, const double* __restrict__ offs,
                                                    cplx* __restrict__ rkT /*[pairs][rows][Mp]*/)
{
    const int pair = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * Mp) return;
    const int r = (int)(i / Mp), u = (int)(i - (long long)r * Mp);
    cplx w = make_double2(0., 0.);
    if (u < up) {
        const double cr = -2.0 * 3.141592653589793 / ((double)rows * uf);
        double ph = cr * (((double)u - offs[2 * pair]) * fftfreq_idx(r, rows));
        double s, co;
        sincos(ph, &s, &co);
        w = make_double2(co, s);
    }
    rkT[((size_t)pair * rows + r) * Mp + u] = w;
}

// grid = (cols / 16, Mp / 32 rounded up, pairs), block = 64
__global__ void __launch_bounds__(64) k6_dft_mfma(const cplx* __restrict__ data, const cplx* __restrict__ rkT, int rows, int cols,
                                                  int Mp, cplx* __restrict__ T /*[pairs][Mp][cols]*/)
{
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    const int c0 = blockIdx.x * 16, m0 = blockIdx.y * 32, pair = blockIdx.z;
    const bool two = (m0 + 16) < Mp;                       // Mp is a multiple of 16: the last block may hold one M-tile
    const cplx* d = data + (size_t)pair * rows * cols + c0 + li;
    const cplx* a = rkT + (size_t)pair * rows * Mp + m0 + li;
    v4f64 re0 = {0., 0., 0., 0.}, im0 = re0, re1 = re0, im1 = re0;
    for (int r0 = 0; r0 < rows; r0 += 4) {
        const int r = r0 + lk;
        const cplx b = d[(size_t)r * cols];
        const cplx a0 = a[(size_t)r * Mp];
        const cplx a1 = two ? a[(size_t)r * Mp + 16] : make_double2(0., 0.);
        re0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b.x, re0, 0, 0, 0);
        re0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0.y, b.y, re0, 0, 0, 0);
        im0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b.y, im0, 0, 0, 0);
        im0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b.x, im0, 0, 0, 0);
        re1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b.x, re1, 0, 0, 0);
        re1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1.y, b.y, re1, 0, 0, 0);
        im1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b.y, im1, 0, 0, 0);
        im1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b.x, im1, 0, 0, 0);
    }
    cplx* t = T + (size_t)pair * Mp * cols + c0 + li;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int u = m0 + lk + 4 * g;
        t[(size_t)u * cols] = make_double2(re0[g], im0[g]);
        if (two) t[(size_t)(u + 16) * cols] = make_double2(re1[g], im1[g]);
    }
}

// out[u][v] = sum_c T[u][c] * ck[c][v]; grid = (up, pairs), block = 256 (same reduction tree as k6_dftups' stage 2)
__global__ void __launch_bounds__(256) k6_dft_cols(const cplx* __restrict__ T, int cols, int Mp, int up, int uf,
                                                   const double* __restrict__ offs, cplx* __restrict__ out)
{
    __shared__ cplx wsum[4];
    const int u = blockIdx.x, pair = blockIdx.y, tid = threadIdx.x;
    const cplx* t = T + ((size_t)pair * Mp + u) * cols;
    const double coff = offs[2 * pair + 1];
    const double cc = -2.0 * 3.141592653589793 / ((double)cols * uf);
    for (int v = 0; v < up; v++) {
        cplx acc = make_double2(0., 0.);
        for (int c = tid; c < cols; c += blockDim.x) {
            double ph = cc * (fftfreq_idx(c, cols) * ((double)v - coff));
            double s, co;
            sincos(ph, &s, &co);
            const cplx x = t[c];
            acc.x += x.x * co - x.y * s;
            acc.y += x.x * s + x.y * co;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); }
        if ((tid & 63) == 0) wsum[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            cplx x = wsum[0];
            for (int k = 1; k < 4; k++) { x.x += wsum[k].x; x.y += wsum[k].y; }
            out[((size_t)pair * up + u) * up + v] = x;
        }
        __syncthreads();
    }
}

// ---- real-transform path: kernels ---------------------------------------------------------------------------------
__global__ void k6_u16_to_f64(const uint16_t* __restrict__ a, double* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)a[i];
}

// Ph = Fh * conj(Gh) on the half spectrum [H][Wh]; per-pair sums of |F|^2, |G|^2 over the FULL spectrum (columns that
// stand for themselves and their mirror image count twice)   (phase_correlate.py:71, 86-87, 117-120)
__global__ void __launch_bounds__(256) k6_cross_power_half(const cplx* __restrict__ F, const cplx* __restrict__ G, cplx* __restrict__ prod,
                                                           int H, int W, int Wh, double* __restrict__ sums /*[pairs][2]*/)
{
    const int pair = blockIdx.y;
    const size_t nh = (size_t)H * Wh;
    const cplx* f = F + (size_t)pair * nh;
    const cplx* g = G + (size_t)pair * nh;
    cplx* p = prod + (size_t)pair * nh;
    double sf = 0., sg = 0.;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nh; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (size_t)Wh);
        const double wgt = (c == 0 || 2 * c == W) ? 1.0 : 2.0;
        const cplx a = f[i], b = g[i];
        p[i] = cmul(a, cconj(b));
        sf += wgt * (a.x * a.x + a.y * a.y);
        sg += wgt * (b.x * b.x + b.y * b.y);
    }
    __shared__ double r0[256], r1[256];
    r0[threadIdx.x] = sf; r1[threadIdx.x] = sg;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[((size_t)pair * gridDim.x + blockIdx.x) * 2] = r0[0]; sums[((size_t)pair * gridDim.x + blockIdx.x) * 2 + 1] = r1[0]; }
}

// argmax of a real array per pair, two levels (parts x pairs blocks, then one block per pair); first maximum wins
__global__ void __launch_bounds__(256) k6_argmax_real_part(const double* __restrict__ data, size_t n, int parts, Peak* __restrict__ part_out)
{
    const int pair = blockIdx.y, part = blockIdx.x;
    const double* d = data + (size_t)pair * n;
    const size_t lo = n * part / parts, hi = n * (part + 1) / parts;
    Peak best;
    best.re = -__builtin_inf(); best.im = 0.0; best.idx = (long long)n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double v = d[i];
        if (v > best.re || (v == best.re && (long long)i < best.idx)) { best.re = v; best.idx = (long long)i; }
    }
    __shared__ Peak sh[256];
    sh[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o && peak_better(sh[threadIdx.x + o], sh[threadIdx.x])) sh[threadIdx.x] = sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part_out[(size_t)pair * parts + part] = sh[0];
}

// Whole-pixel stage on the device (phase_correlate.py:73-107): reduce the partial peaks, turn the peak into a shift,
// and either finish (upsample_factor == 1: error by Parseval, :85-92) or emit the offsets of the upsampled DFT.
__global__ void k6_coarse(const Peak* __restrict__ part, int parts, int n_pairs, int H, int W, int uf,
                          const double* __restrict__ psums /*[pairs][nsum][2]*/, int nsum, double* __restrict__ sums /*[pairs][2]*/,
                          double* __restrict__ shifts /*[pairs][2]*/, double* __restrict__ offs /*[pairs][2]*/, double* __restrict__ out4)
{
    // one 64-lane block per pair: the partial sums are added up in a fixed order (strided per lane, then a shuffle tree)
    const int p = blockIdx.x;
    {
        double sf = 0.0, sg = 0.0;
        for (int k = threadIdx.x; k < nsum; k += 64) { sf += psums[((size_t)p * nsum + k) * 2]; sg += psums[((size_t)p * nsum + k) * 2 + 1]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sf += __shfl_xor(sf, o); sg += __shfl_xor(sg, o); }
        if (threadIdx.x != 0) return;
        sums[2 * p] = sf; sums[2 * p + 1] = sg;
    }
    Peak best = part[(size_t)p * parts];
    for (int k = 1; k < parts; k++) if (peak_better(part[(size_t)p * parts + k], best)) best = part[(size_t)p * parts + k];
    const double npix = (double)H * (double)W;
    const double mid_row = trunc(H / 2.0), mid_col = trunc(W / 2.0);
    const double row_max = (double)(best.idx / W), col_max = (double)(best.idx % W);
    double rs = row_max > mid_row ? row_max - H : row_max;
    double cs = col_max > mid_col ? col_max - W : col_max;
    if (uf == 1) {
        const double rf = sums[2 * p] / npix, rg = sums[2 * p + 1] / npix;
        const double re = best.re / npix, im = best.im / npix;     // (unnormalised inverse transform; im = 0 on the real path)
        const double err = 1.0 - (re * re + im * im) / (rg * rf);
        out4[4 * p] = rs; out4[4 * p + 1] = cs;
        out4[4 * p + 2] = sqrt(fabs(err)); out4[4 * p + 3] = atan2(im, re);
        return;
    }
    const double dftshift = trunc(ceil(uf * 1.5) / 2.0);
    rs = nearbyint(rs * uf) / uf;                           // numpy.round: half to even
    cs = nearbyint(cs * uf) / uf;
    shifts[2 * p] = rs; shifts[2 * p + 1] = cs;
    offs[2 * p] = dftshift - rs * uf; offs[2 * p + 1] = dftshift - cs * uf;
}

// Sub-pixel stage (phase_correlate.py:109-134): peak of the upsampled cross-correlation -> shift, error, diffphase.
__global__ void k6_fine(const Peak* __restrict__ peaks, int n_pairs, int H, int W, int uf, int up, const double* __restrict__ sums,
                        const double* __restrict__ shifts, double* __restrict__ out4)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const double mid_row = trunc(H / 2.0), mid_col = trunc(W / 2.0);
    const double dftshift = trunc(up / 2.0);
    const double norm = mid_row * mid_col * (double)uf * uf;
    const Peak pk = peaks[p];
    const double rm = (double)(pk.idx / up) - dftshift, cm = (double)(pk.idx % up) - dftshift;
    double rs = shifts[2 * p] + rm / uf, cs = shifts[2 * p + 1] + cm / uf;
    const double rg00 = sums[2 * p] / norm, rf00 = sums[2 * p + 1] / norm;
    const double err = 1.0 - (pk.re * pk.re + pk.im * pk.im) / (rg00 * rf00);
    if (mid_row == 1) rs = 0;
    if (mid_col == 1) cs = 0;
    out4[4 * p] = rs; out4[4 * p + 1] = cs; out4[4 * p + 2] = sqrt(fabs(err)); out4[4 * p + 3] = atan2(pk.im, pk.re);
}

// data[r][c] = G[r][c] * conj(F[r][c]) (phase_correlate.py:102) from the two half spectra; columns beyond W/2 by Hermitian
// symmetry of the spectra of real images: X[r][c] = conj(X[(H - r) % H][W - c])
__device__ __forceinline__ cplx gf_full(const cplx* __restrict__ Fh, const cplx* __restrict__ Gh, int H, int W, int Wh, int r, int c)
{
    if (c < Wh) {
        const size_t i = (size_t)r * Wh + c;
        return cmul(Gh[i], cconj(Fh[i]));
    }
    const size_t i = (size_t)((H - r) % H) * Wh + (W - c);
    return cmul(cconj(Gh[i]), Fh[i]);
}

// The matrix-core row product with its B operand formed on the fly from the half spectra (see k6_dft_mfma).
__global__ void __launch_bounds__(64) k6_dft_mfma_half(const cplx* __restrict__ Fh, const cplx* __restrict__ Gh, const cplx* __restrict__ rkT,
                                                       int rows, int cols, int Wh, int Mp, cplx* __restrict__ T /*[pairs][Mp][cols]*/)
{
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    const int c0 = blockIdx.x * 16, m0 = blockIdx.y * 32, pair = blockIdx.z;
    const bool two = (m0 + 16) < Mp;
    const cplx* fh = Fh + (size_t)pair * rows * Wh;
    const cplx* gh = Gh + (size_t)pair * rows * Wh;
    const cplx* a = rkT + (size_t)pair * rows * Mp + m0 + li;
    const int c = c0 + li;
    v4f64 re0 = {0., 0., 0., 0.}, im0 = re0, re1 = re0, im1 = re0;
    for (int r0 = 0; r0 < rows; r0 += 4) {
        const int r = r0 + lk;
        const cplx b = gf_full(fh, gh, rows, cols, Wh, r, c);
        const cplx a0 = a[(size_t)r * Mp];
        const cplx a1 = two ? a[(size_t)r * Mp + 16] : make_double2(0., 0.);
        re0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b.x, re0, 0, 0, 0);
        re0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0.y, b.y, re0, 0, 0, 0);
        im0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b.y, im0, 0, 0, 0);
        im0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b.x, im0, 0, 0, 0);
        re1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b.x, re1, 0, 0, 0);
        re1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1.y, b.y, re1, 0, 0, 0);
        im1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b.y, im1, 0, 0, 0);
        im1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b.x, im1, 0, 0, 0);
    }
    cplx* t = T + (size_t)pair * Mp * cols + c;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int u = m0 + lk + 4 * g;
        t[(size_t)u * cols] = make_double2(re0[g], im0[g]);
        if (two) t[(size_t)(u + 16) * cols] = make_double2(re1[g], im1[g]);
    }
}

// full-spectrum G * conj(F) from the half spectra, for the vector-ALU DFT (shapes the MFMA tiles do not fit)
__global__ void k6_expand_gf(const cplx* __restrict__ Fh, const cplx* __restrict__ Gh, int H, int W, int Wh, cplx* __restrict__ out)
{
    const int pair = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const int r = (int)(i / W), c = (int)(i % W);
    out[(size_t)pair * H * W + i] = gf_full(Fh + (size_t)pair * H * Wh, Gh + (size_t)pair * H * Wh, H, W, Wh, r, c);
}

// ---- plans: one per (shape, batch, kind, device, stream), work areas supplied by the caller ------------------------
enum PlanKind { PK_Z2Z = 0, PK_D2Z = 1, PK_Z2D = 2 };
struct PlanKey {
    int H, W, batch, kind, dev; void* stream;
    bool operator<(const PlanKey& o) const { return std::tie(H, W, batch, kind, dev, stream) < std::tie(o.H, o.W, o.batch, o.kind, o.dev, o.stream); }
};
struct PlanEntry { hipfftHandle h; size_t work; unsigned long long stamp; };
std::mutex g_plan_mu;                   // held from the lookup to the last exec of a call: a plan (its stream, its work
std::map<PlanKey, PlanEntry> g_plans;   // area) is never used by two host threads at once
unsigned long long g_plan_clock = 0;
constexpr size_t MAX_PLANS = 24;

// (g_plan_mu held)
int get_plan(int H, int W, int batch, int kind, hipStream_t s, PlanEntry* out)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    PlanKey k{H, W, batch, kind, dev, (void*)s};
    auto it = g_plans.find(k);
    if (it != g_plans.end()) { it->second.stamp = ++g_plan_clock; *out = it->second; return FSQ_OK; }
    if (g_plans.size() >= MAX_PLANS) {          // least recently used plan goes (offsets_from_frames with varying frame counts)
        auto old = g_plans.begin();
        for (auto j = g_plans.begin(); j != g_plans.end(); ++j) if (j->second.stamp < old->second.stamp) old = j;
        (void)hipfftDestroy(old->second.h);
        g_plans.erase(old);
    }
    hipfftHandle h;
    if (hipfftCreate(&h) != HIPFFT_SUCCESS) return FSQ_EHIP;
    int n[2] = {H, W};
    size_t work = 0;
    const hipfftType ty = kind == PK_Z2Z ? HIPFFT_Z2Z : kind == PK_D2Z ? HIPFFT_D2Z : HIPFFT_Z2D;
    if (hipfftSetAutoAllocation(h, 0) != HIPFFT_SUCCESS ||
        hipfftMakePlanMany(h, 2, n, nullptr, 1, 0, nullptr, 1, 0, ty, batch, &work) != HIPFFT_SUCCESS ||
        hipfftSetStream(h, s) != HIPFFT_SUCCESS) {
        (void)hipfftDestroy(h);
        return FSQ_ENOTIMPL;
    }
    PlanEntry e{h, work, ++g_plan_clock};
    g_plans[k] = e;
    *out = e;
    return FSQ_OK;
}

struct RegLayout {
    size_t A, B, Fh, Gh, Ph, cc, T, rkT, Tg, U, psums, sums, shifts, offs, part, peaks, fftwork, total;
    int parts, Wh, up, Mp; bool real_path, mfma;
};

size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

RegLayout reg_layout(int n_pairs, int H, int W, int uf, int dtype, bool real_path, size_t fft_work)
{
    RegLayout L;
    const size_t npix = (size_t)H * W;
    L.real_path = real_path;
    L.Wh = W / 2 + 1;
    L.up = (int)ceil(uf * 1.5);
    L.Mp = 16 * ((L.up + 15) / 16);
    L.mfma = (W % 16) == 0 && (H % 4) == 0 && L.Mp <= W && L.Mp <= H;
    L.parts = 16;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += al(bytes); return at; };
    if (real_path) {
        const size_t nh = (size_t)H * L.Wh;
        L.A = take(dtype == FSQ_DTYPE_U16 ? n_pairs * npix * 8 : 0);
        L.B = take(dtype == FSQ_DTYPE_U16 ? n_pairs * npix * 8 : 0);
        L.Fh = take(n_pairs * nh * 16); L.Gh = take(n_pairs * nh * 16); L.Ph = take(n_pairs * nh * 16);
        L.cc = take(n_pairs * npix * 8);
        // upsampled DFT: T (Mp x W) and rkT (H x Mp) per pair, or the expanded full spectrum for the vector-ALU path
        L.T = take(uf > 1 ? (L.mfma ? (size_t)n_pairs * L.Mp * W * 16 : (size_t)n_pairs * npix * 16) : 0);
        L.rkT = take(uf > 1 && L.mfma ? (size_t)n_pairs * H * L.Mp * 16 : 0);
    } else {
        L.A = L.B = 0;
        L.Fh = take(n_pairs * npix * 16); L.Gh = take(n_pairs * npix * 16); L.Ph = take(n_pairs * npix * 16);
        L.cc = L.T = L.rkT = 0;             // (the full-spectrum path re-uses F and G for rkT and T)
    }
    L.Tg = take(uf > 1 && !L.mfma && W > DFT_TC ? (size_t)n_pairs * L.up * W * 16 : 0);     // vector-ALU DFT: T rows too long for LDS
    L.U = take(uf > 1 ? (size_t)n_pairs * L.up * L.up * 16 : 0);
    L.psums = take((size_t)n_pairs * 1024 * 16);
    L.sums = take((size_t)n_pairs * 16); L.shifts = take((size_t)n_pairs * 16); L.offs = take((size_t)n_pairs * 16);
    L.part = take((size_t)n_pairs * L.parts * sizeof(Peak)); L.peaks = take((size_t)n_pairs * sizeof(Peak));
    L.fftwork = take(fft_work);
    L.total = o;
    return L;
}

bool want_z2z() { const char* e = getenv("FSQ_REGISTER_Z2Z"); return e && atoi(e); }

// plans of one call + the work area they need (g_plan_mu held)
int reg_plans(int n_pairs, int H, int W, hipStream_t s, bool* real_path, PlanEntry* fwd, PlanEntry* inv, size_t* work)
{
    *real_path = !want_z2z();
    if (*real_path) {
        int rc = get_plan(H, W, n_pairs, PK_D2Z, s, fwd);
        if (rc == FSQ_OK) rc = get_plan(H, W, n_pairs, PK_Z2D, s, inv);
        if (rc == FSQ_OK) { *work = fwd->work > inv->work ? fwd->work : inv->work; return FSQ_OK; }
        *real_path = false;                 // no real-transform plan for this shape: full-spectrum path
    }
    int rc = get_plan(H, W, n_pairs, PK_Z2Z, s, fwd);
    if (rc != FSQ_OK) return rc;
    *inv = *fwd; *work = fwd->work;
    return FSQ_OK;
}

}  // namespace

extern "C" int64_t fsq_phase_correlate_workspace_bytes(int n_pairs, int H, int W, int upsample_factor, int dtype, void* stream)
{
    if (n_pairs < 1 || H < 1 || W < 1 || upsample_factor < 1 || (dtype != FSQ_DTYPE_F64 && dtype != FSQ_DTYPE_U16)) return FSQ_EINVAL;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    bool real_path; PlanEntry fwd, inv; size_t work = 0;
    int rc = reg_plans(n_pairs, H, W, (hipStream_t)stream, &real_path, &fwd, &inv, &work);
    if (rc != FSQ_OK) return rc;
    return (int64_t)reg_layout(n_pairs, H, W, upsample_factor, dtype, real_path, work).total;
}

extern "C" int fsq_phase_correlate(const void* d_ref, const void* d_reg, int dtype, int n_pairs, int H, int W, int upsample_factor,
                                   double* d_out4, void* d_workspace, int64_t workspace_bytes, void* stream)
{
    if (n_pairs < 1 || H < 1 || W < 1 || upsample_factor < 1 || !d_ref || !d_reg || !d_out4 || !d_workspace) return FSQ_EINVAL;
    if (dtype != FSQ_DTYPE_F64 && dtype != FSQ_DTYPE_U16) return FSQ_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t npix = (size_t)H * W, ntot = npix * n_pairs;
    const int uf = upsample_factor;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    bool real_path; PlanEntry fwd, inv; size_t work = 0;
    int rc = reg_plans(n_pairs, H, W, s, &real_path, &fwd, &inv, &work);
    if (rc != FSQ_OK) return rc;
    const RegLayout L = reg_layout(n_pairs, H, W, uf, dtype, real_path, work);
    if ((size_t)workspace_bytes < L.total) return FSQ_ENOMEM;
    unsigned char* ws = (unsigned char*)d_workspace;
    double* sums = (double*)(ws + L.sums);
    double* psums = (double*)(ws + L.psums);
    double* shifts = (double*)(ws + L.shifts);
    double* offs = (double*)(ws + L.offs);
    Peak* part = (Peak*)(ws + L.part);
    Peak* peaks = (Peak*)(ws + L.peaks);
    cplx* U = (cplx*)(ws + L.U);
    const int up = L.up, Mp = L.Mp;
    const unsigned pb = (unsigned)((n_pairs + 63) / 64);
    const bool use_mfma = L.mfma && !getenv("FSQ_REGISTER_NO_MFMA");
    if (hipfftSetWorkArea(fwd.h, ws + L.fftwork) != HIPFFT_SUCCESS) return FSQ_EHIP;
    if (inv.h != fwd.h && hipfftSetWorkArea(inv.h, ws + L.fftwork) != HIPFFT_SUCCESS) return FSQ_EHIP;
    const unsigned blocks = (unsigned)((ntot + 255) / 256);
    if (real_path) {
        const int Wh = L.Wh;
        const size_t nh = (size_t)H * Wh;
        const double *A = (const double*)d_ref, *B = (const double*)d_reg;
        if (dtype == FSQ_DTYPE_U16) {
            hipLaunchKernelGGL(k6_u16_to_f64, dim3(blocks), dim3(256), 0, s, (const uint16_t*)d_ref, (double*)(ws + L.A), ntot);
            hipLaunchKernelGGL(k6_u16_to_f64, dim3(blocks), dim3(256), 0, s, (const uint16_t*)d_reg, (double*)(ws + L.B), ntot);
            A = (const double*)(ws + L.A); B = (const double*)(ws + L.B);
        }
        cplx *Fh = (cplx*)(ws + L.Fh), *Gh = (cplx*)(ws + L.Gh), *Ph = (cplx*)(ws + L.Ph);
        double* cc = (double*)(ws + L.cc);
        if (hipfftExecD2Z(fwd.h, (hipfftDoubleReal*)A, (hipfftDoubleComplex*)Fh) != HIPFFT_SUCCESS) return FSQ_EHIP;
        if (hipfftExecD2Z(fwd.h, (hipfftDoubleReal*)B, (hipfftDoubleComplex*)Gh) != HIPFFT_SUCCESS) return FSQ_EHIP;
        unsigned gx = (unsigned)((nh + 255) / 256);
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(k6_cross_power_half, dim3(gx, n_pairs), dim3(256), 0, s, Fh, Gh, Ph, H, W, Wh, psums);
        if (hipfftExecZ2D(inv.h, (hipfftDoubleComplex*)Ph, (hipfftDoubleReal*)cc) != HIPFFT_SUCCESS) return FSQ_EHIP;
        hipLaunchKernelGGL(k6_argmax_real_part, dim3(L.parts, n_pairs), dim3(256), 0, s, cc, npix, L.parts, part);
        hipLaunchKernelGGL(k6_coarse, dim3(n_pairs), dim3(64), 0, s, part, L.parts, n_pairs, H, W, uf, psums, (int)gx, sums, shifts, offs, d_out4);
        if (uf > 1) {
            cplx* T = (cplx*)(ws + L.T);
            if (use_mfma) {
                cplx* rkT = (cplx*)(ws + L.rkT);
                hipLaunchKernelGGL(k6_rowkernel, dim3((unsigned)(((size_t)H * Mp + 255) / 256), n_pairs), dim3(256), 0, s, H, up, Mp, uf, offs, rkT);
                hipLaunchKernelGGL(k6_dft_mfma_half, dim3(W / 16, (Mp + 31) / 32, n_pairs), dim3(64), 0, s, Fh, Gh, rkT, H, W, Wh, Mp, T);
                hipLaunchKernelGGL(k6_dft_cols, dim3(up, n_pairs), dim3(256), 0, s, T, W, Mp, up, uf, offs, U);
            } else {
                hipLaunchKernelGGL(k6_expand_gf, dim3((unsigned)((npix + 255) / 256), n_pairs), dim3(256), 0, s, Fh, Gh, H, W, Wh, T);
                hipLaunchKernelGGL(k6_dftups, dim3(up, n_pairs), dim3(256), (size_t)(DFT_RC + DFT_TC) * sizeof(cplx), s, T, H, W, up, uf, offs, U,
                                   (cplx*)(ws + L.Tg));
            }
        }
    } else {
        cplx *F = (cplx*)(ws + L.Fh), *G = (cplx*)(ws + L.Gh), *P = (cplx*)(ws + L.Ph);
        if (dtype == FSQ_DTYPE_U16) {
            hipLaunchKernelGGL(k6_u16_to_complex, dim3(blocks), dim3(256), 0, s, (const uint16_t*)d_ref, F, ntot);
            hipLaunchKernelGGL(k6_u16_to_complex, dim3(blocks), dim3(256), 0, s, (const uint16_t*)d_reg, G, ntot);
        } else {
            hipLaunchKernelGGL(k6_to_complex, dim3(blocks), dim3(256), 0, s, (const double*)d_ref, F, ntot);
            hipLaunchKernelGGL(k6_to_complex, dim3(blocks), dim3(256), 0, s, (const double*)d_reg, G, ntot);
        }
        if (hipfftExecZ2Z(fwd.h, (hipfftDoubleComplex*)F, (hipfftDoubleComplex*)F, HIPFFT_FORWARD) != HIPFFT_SUCCESS) return FSQ_EHIP;
        if (hipfftExecZ2Z(fwd.h, (hipfftDoubleComplex*)G, (hipfftDoubleComplex*)G, HIPFFT_FORWARD) != HIPFFT_SUCCESS) return FSQ_EHIP;
        unsigned gx = (unsigned)((npix + 255) / 256);
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(k6_cross_power, dim3(gx, n_pairs), dim3(256), 0, s, F, G, P, npix, psums);
        if (hipfftExecZ2Z(fwd.h, (hipfftDoubleComplex*)P, (hipfftDoubleComplex*)P, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) return FSQ_EHIP;
        hipLaunchKernelGGL(k6_argmax, dim3(n_pairs), dim3(256), 0, s, P, npix, 1.0, 0, part);     // parts = 1 layout below
        // (k6_coarse reads `parts` entries per pair: hand it the single complex peak with parts = 1; imaginary part kept)
        hipLaunchKernelGGL(k6_coarse, dim3(n_pairs), dim3(64), 0, s, part, 1, n_pairs, H, W, uf, psums, (int)gx, sums, shifts, offs, d_out4);
        if (uf > 1) {
            hipLaunchKernelGGL(k6_cross_power2, dim3(blocks), dim3(256), 0, s, F, G, P, ntot);
            if (use_mfma) {
                cplx* rkT = F;          // F and G are free by now; each pair's slot holds rows x Mp resp. Mp x cols elements
                cplx* Tm = G;
                hipLaunchKernelGGL(k6_rowkernel, dim3((unsigned)(((size_t)H * Mp + 255) / 256), n_pairs), dim3(256), 0, s, H, up, Mp, uf, offs, rkT);
                hipLaunchKernelGGL(k6_dft_mfma, dim3(W / 16, (Mp + 31) / 32, n_pairs), dim3(64), 0, s, P, rkT, H, W, Mp, Tm);
                hipLaunchKernelGGL(k6_dft_cols, dim3(up, n_pairs), dim3(256), 0, s, Tm, W, Mp, up, uf, offs, U);
            } else {
                hipLaunchKernelGGL(k6_dftups, dim3(up, n_pairs), dim3(256), (size_t)(DFT_RC + DFT_TC) * sizeof(cplx), s, P, H, W, up, uf, offs, U,
                                   (cplx*)(ws + L.Tg));
            }
        }
    }
    if (uf > 1) {
        const double norm = trunc(H / 2.0) * trunc(W / 2.0) * (double)uf * uf;
        hipLaunchKernelGGL(k6_argmax, dim3(n_pairs), dim3(256), 0, s, U, (size_t)up * up, 1.0 / norm, 1, peaks);
        hipLaunchKernelGGL(k6_fine, dim3(pb), dim3(64), 0, s, peaks, n_pairs, H, W, uf, up, sums, shifts, d_out4);
    }
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}
