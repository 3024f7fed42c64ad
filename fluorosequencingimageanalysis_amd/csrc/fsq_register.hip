// fsq_register.hip - K6: sub-pixel frame registration by phase correlation, batched over image pairs.
// Reference: phase_correlate.phase_correlate / _dftups, phase_correlate.py:11-196 (Guizar-Sicairos et al.).
//
//   F = fft2(ref), G = fft2(reg)                       rocFFT (hipFFT front end), complex128, batched
//   cc = ifft2(F * conj(G))                             cross-power spectrum kernel + inverse FFT
//   (row_max, col_max) = argmax(cc)                     lexicographic (real, imag) like numpy, first hit
//   upsample_factor > 1: matrix-multiply DFT of G*conj(F) on a ceil(1.5*uf)^2 grid around the peak,
//                        argmax again, error / diffphase from the peak value       (:94-122)
// All spectra stay in HBM (3 x 16 B/px per pair); the kernels are HBM-streaming (cross-power, reductions) except
// the upsampled DFT, whose row product is a complex GEMM on v_mfma_f64_16x16x4_f64 (k6_dft_mfma below);
// everything in fp64 like the reference.
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
    if (threadIdx.x == 0) { atomicAdd(&sums[2 * pair], r0[0]); atomicAdd(&sums[2 * pair + 1], r1[0]); }
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
// grid = (up, pairs); block = 256; dynamic LDS = (rows + cols) complex.
//   stage 1: the row kernel rk[u][.] of this block's u is evaluated ONCE into LDS (rows sincos instead of rows x cols),
//            then T[c] = sum_r rk[u][r] * data[r][c], thread per column, r in order (coalesced reads of data);
//   stage 2: out[u][v] = sum_c T[c] * ck[c][v]: every thread takes the columns c = tid, tid + 256, ..., the partial sums
//            are combined by a fixed-shape tree (wave shuffles, then the 4 waves in order) - deterministic.
__global__ void __launch_bounds__(256) k6_dftups(const cplx* __restrict__ data, int rows, int cols, int up, int uf,
                                                 const double* __restrict__ offs /*[pairs][2] row_off, col_off*/,
                                                 cplx* __restrict__ out /*[pairs][up][up]*/)
{
    extern __shared__ cplx lds_c[];
    cplx* Wr = lds_c;                                 // [rows]
    cplx* T = lds_c + rows;                           // [cols]
    __shared__ cplx wsum[4];
    const int u = blockIdx.x, pair = blockIdx.y, tid = threadIdx.x;
    const cplx* d = data + (size_t)pair * rows * cols;
    const double roff = offs[2 * pair], coff = offs[2 * pair + 1];
    const double cr = -2.0 * 3.141592653589793 / ((double)rows * uf), cc = -2.0 * 3.141592653589793 / ((double)cols * uf);
    for (int r = tid; r < rows; r += blockDim.x) {
        double ph = cr * (((double)u - roff) * fftfreq_idx(r, rows));
        double s, co;
        sincos(ph, &s, &co);
        Wr[r] = make_double2(co, s);
    }
    __syncthreads();
    for (int c = tid; c < cols; c += blockDim.x) {
        cplx acc = make_double2(0., 0.);
        for (int r = 0; r < rows; r++) {
            const cplx w = Wr[r];
            const cplx v = d[(size_t)r * cols + c];
            acc.x += w.x * v.x - w.y * v.y;
            acc.y += w.x * v.y + w.y * v.x;
        }
        T[c] = acc;
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

struct PlanKey {
    int H, W, batch, dev;
    bool operator<(const PlanKey& o) const { return std::tie(H, W, batch, dev) < std::tie(o.H, o.W, o.batch, o.dev); }
};
std::mutex g_plan_mu;
std::map<PlanKey, hipfftHandle> g_plans;

int get_plan(int H, int W, int batch, hipfftHandle* out)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    PlanKey k{H, W, batch, dev};
    auto it = g_plans.find(k);
    if (it != g_plans.end()) { *out = it->second; return FSQ_OK; }
    hipfftHandle h;
    int n[2] = {H, W};
    if (hipfftPlanMany(&h, 2, n, nullptr, 1, H * W, nullptr, 1, H * W, HIPFFT_Z2Z, batch) != HIPFFT_SUCCESS) return FSQ_EHIP;
    g_plans[k] = h;
    *out = h;
    return FSQ_OK;
}

}  // namespace

extern "C" int fsq_phase_correlate(const double* d_ref, const double* d_reg, int n_pairs, int H, int W, int upsample_factor,
                                   double* d_out4, void* stream)
{
    if (n_pairs < 1 || H < 1 || W < 1 || upsample_factor < 1 || !d_ref || !d_reg || !d_out4) return FSQ_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t npix = (size_t)H * W, ntot = npix * n_pairs;
    const int uf = upsample_factor;
    const int up = (int)ceil(uf * 1.5);
    cplx *F = nullptr, *G = nullptr, *P = nullptr, *U = nullptr;
    double *sums = nullptr, *offs = nullptr;
    Peak* peaks = nullptr;
    int rc = FSQ_OK;
    std::vector<Peak> hp(n_pairs), hp2(n_pairs);
    std::vector<double> hs(2 * n_pairs), hoff(2 * n_pairs), hout(4 * n_pairs);
    std::vector<double> rshift(n_pairs), cshift(n_pairs);
    const double mid_row = trunc(H / 2.0), mid_col = trunc(W / 2.0);
    hipfftHandle plan;
#define CK(e) do { if ((e) != hipSuccess) { g_fsq_last_hip = (e); rc = FSQ_EHIP; goto done; } } while (0)
    CK(hipMallocAsync((void**)&F, ntot * sizeof(cplx), s));
    CK(hipMallocAsync((void**)&G, ntot * sizeof(cplx), s));
    CK(hipMallocAsync((void**)&P, ntot * sizeof(cplx), s));
    CK(hipMallocAsync((void**)&sums, 2 * n_pairs * sizeof(double), s));
    CK(hipMallocAsync((void**)&offs, 2 * n_pairs * sizeof(double), s));
    CK(hipMallocAsync((void**)&peaks, n_pairs * sizeof(Peak), s));
    if (uf > 1) CK(hipMallocAsync((void**)&U, (size_t)n_pairs * up * up * sizeof(cplx), s));
    CK(hipMemsetAsync(sums, 0, 2 * n_pairs * sizeof(double), s));
    rc = get_plan(H, W, n_pairs, &plan);
    if (rc != FSQ_OK) goto done;
    if (hipfftSetStream(plan, s) != HIPFFT_SUCCESS) { rc = FSQ_EHIP; goto done; }
    {
        const unsigned blocks = (unsigned)((ntot + 255) / 256);
        hipLaunchKernelGGL(k6_to_complex, dim3(blocks), dim3(256), 0, s, d_ref, F, ntot);
        hipLaunchKernelGGL(k6_to_complex, dim3(blocks), dim3(256), 0, s, d_reg, G, ntot);
        if (hipfftExecZ2Z(plan, (hipfftDoubleComplex*)F, (hipfftDoubleComplex*)F, HIPFFT_FORWARD) != HIPFFT_SUCCESS) { rc = FSQ_EHIP; goto done; }
        if (hipfftExecZ2Z(plan, (hipfftDoubleComplex*)G, (hipfftDoubleComplex*)G, HIPFFT_FORWARD) != HIPFFT_SUCCESS) { rc = FSQ_EHIP; goto done; }
        unsigned gx = (unsigned)((npix + 255) / 256);
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(k6_cross_power, dim3(gx, n_pairs), dim3(256), 0, s, F, G, P, npix, sums);
        if (hipfftExecZ2Z(plan, (hipfftDoubleComplex*)P, (hipfftDoubleComplex*)P, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) { rc = FSQ_EHIP; goto done; }
        hipLaunchKernelGGL(k6_argmax, dim3(n_pairs), dim3(256), 0, s, P, npix, 1.0 / (double)npix, 0, peaks);
    }
    CK(hipMemcpyAsync(hp.data(), peaks, n_pairs * sizeof(Peak), hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(hs.data(), sums, 2 * n_pairs * sizeof(double), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    for (int p = 0; p < n_pairs; p++) {                                   // phase_correlate.py:73-84
        double row_max = (double)(hp[p].idx / W), col_max = (double)(hp[p].idx % W);
        rshift[p] = row_max > mid_row ? row_max - H : row_max;
        cshift[p] = col_max > mid_col ? col_max - W : col_max;
    }
    if (uf == 1) {
        for (int p = 0; p < n_pairs; p++) {                               // :85-92
            double rf = hs[2 * p] / (double)npix, rg = hs[2 * p + 1] / (double)npix;
            double re = hp[p].re, im = hp[p].im;
            double err = 1.0 - (re * re + im * im) / (rg * rf);
            hout[4 * p] = rshift[p]; hout[4 * p + 1] = cshift[p];
            hout[4 * p + 2] = sqrt(fabs(err)); hout[4 * p + 3] = atan2(im, re);
        }
    } else {
        const double dftshift = trunc(up / 2.0);
        for (int p = 0; p < n_pairs; p++) {                               // :96-107
            rshift[p] = nearbyint(rshift[p] * uf) / uf;
            cshift[p] = nearbyint(cshift[p] * uf) / uf;
            hoff[2 * p] = dftshift - rshift[p] * uf;
            hoff[2 * p + 1] = dftshift - cshift[p] * uf;
        }
        CK(hipMemcpyAsync(offs, hoff.data(), 2 * n_pairs * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k6_cross_power2, dim3((unsigned)((ntot + 255) / 256)), dim3(256), 0, s, F, G, P, ntot);
        const int Mp = 16 * ((up + 15) / 16);
        if ((W % 16) == 0 && (H % 4) == 0 && Mp <= W && Mp <= H && !getenv("FSQ_REGISTER_NO_MFMA")) {
            // matrix-core path; F and G are free by now and each pair's slot holds rows x Mp resp. Mp x cols elements
            cplx* rkT = F;
            cplx* Tm = G;
            hipLaunchKernelGGL(k6_rowkernel, dim3((unsigned)(((size_t)H * Mp + 255) / 256), n_pairs), dim3(256), 0, s, H, up, Mp, uf, offs, rkT);
            // (pair slots of rkT / T are packed back to back: H*Mp and Mp*W elements per pair, both <= H*W)
            hipLaunchKernelGGL(k6_dft_mfma, dim3(W / 16, (Mp + 31) / 32, n_pairs), dim3(64), 0, s, P, rkT, H, W, Mp, Tm);
            hipLaunchKernelGGL(k6_dft_cols, dim3(up, n_pairs), dim3(256), 0, s, Tm, W, Mp, up, uf, offs, U);
        } else {
            hipLaunchKernelGGL(k6_dftups, dim3(up, n_pairs), dim3(256), (size_t)(H + W) * sizeof(cplx), s, P, H, W, up, uf, offs, U);
        }
        const double norm = mid_row * mid_col * (double)uf * uf;
        hipLaunchKernelGGL(k6_argmax, dim3(n_pairs), dim3(256), 0, s, U, (size_t)up * up, 1.0 / norm, 1, peaks);
        CK(hipMemcpyAsync(hp2.data(), peaks, n_pairs * sizeof(Peak), hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        for (int p = 0; p < n_pairs; p++) {                               // :109-128
            double rm = (double)(hp2[p].idx / up) - dftshift, cm = (double)(hp2[p].idx % up) - dftshift;
            double rs = rshift[p] + rm / uf, cs = cshift[p] + cm / uf;
            double rg00 = hs[2 * p] / norm, rf00 = hs[2 * p + 1] / norm;
            double re = hp2[p].re, im = hp2[p].im;
            double err = 1.0 - (re * re + im * im) / (rg00 * rf00);
            if (mid_row == 1) rs = 0;
            if (mid_col == 1) cs = 0;
            hout[4 * p] = rs; hout[4 * p + 1] = cs; hout[4 * p + 2] = sqrt(fabs(err)); hout[4 * p + 3] = atan2(im, re);
        }
    }
    CK(hipMemcpyAsync(d_out4, hout.data(), 4 * n_pairs * sizeof(double), hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
done:
    if (F) (void)hipFreeAsync(F, s);
    if (G) (void)hipFreeAsync(G, s);
    if (P) (void)hipFreeAsync(P, s);
    if (U) (void)hipFreeAsync(U, s);
    if (sums) (void)hipFreeAsync(sums, s);
    if (offs) (void)hipFreeAsync(offs, s);
    if (peaks) (void)hipFreeAsync(peaks, s);
#undef CK
    return rc;
}
