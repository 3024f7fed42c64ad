// fsq_fit.hip - K3/K4: per-candidate LM PSF fit + fit-quality metrics, one lane per candidate.
// Reference: the candidate loop of pflib.find_peptides, pflib.py:441-477.
#include <atomic>

#include "fsq_common.h"
#include "fsq_lm_core.h"

#ifdef FSQ_BUILD_AB      // the one-lane-per-fit persistent engine: an A/B build only (make AB=1), not in the shipped library
// pflib.illumina_s_n (pflib.py:261-281) on a 5x5 ROI held as doubles
FSQ_DEV double fsq_illumina_s_n(const double* s, double vmax)
{
    double op[16];
    int t = 0;
    for (int w = 0; w < 5; w++) op[t++] = s[w];
    for (int w = 0; w < 5; w++) op[t++] = s[20 + w];
    for (int h = 1; h < 4; h++) { op[t++] = s[h * 5]; op[t++] = s[h * 5 + 4]; }
    double isum = 0.0;                       // exact: integers < 2^53
    for (int i = 0; i < 16; i++) isum += op[i];
    double mean = isum / 16.0;
    double r[8];
    for (int k = 0; k < 8; k++) {            // numpy pairwise sum of 16 squared deviations
        double d0 = op[k] - mean, d1 = op[8 + k] - mean;
        r[k] = d0 * d0 + d1 * d1;
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    res = 0.0 + res;
    double sd = fsq_sqrt(res / 16.0);
    return (vmax - mean) / sd;
}

// Work queue heads for the persistent fit kernels (one slot per in-flight launch, reset on the stream).
__device__ unsigned long long g_fit_queue[256];

FSQ_DEV void fsq_roi_stats(const double* data, double* vmedian, double* vmax, double* vmean)
{
    double srt[FSQ_NPIX];
    double mx = data[0], isum = 0.0;
    for (int i = 0; i < FSQ_NPIX; i++) { srt[i] = data[i]; mx = data[i] > mx ? data[i] : mx; isum += data[i]; }
    for (int i = 1; i < FSQ_NPIX; i++) {
        double v = srt[i];
        int j = i - 1;
        while (j >= 0 && srt[j] > v) { srt[j + 1] = srt[j]; j--; }
        srt[j + 1] = v;
    }
    *vmedian = srt[12]; *vmax = mx; *vmean = isum / 25.0;
}

// fit-quality metrics and the output row (pflib.py:461-475)
FSQ_DEV void fsq_finish_row(const double* data, const FsqLmState& st, int status, double vmax, double vmean,
                            int h, int w, int field, FsqRow* row)
{
    double fit[FSQ_NPIX];
    fsq_model(st.x, fit);
    double num = 0.0, den = 0.0, rm = 0.0;
    for (int i = 0; i < FSQ_NPIX; i++) { double d = data[i] - fit[i]; num += d * d; }
    for (int i = 0; i < FSQ_NPIX; i++) { double d = data[i] - vmean; den += d * d; }
    for (int i = 0; i < FSQ_NPIX; i++) rm += fsq_pow2(data[i] - fit[i]);
    row->h0 = st.x[2] + h - 2.5;
    row->w0 = st.x[3] + w - 2.5;
    row->H = st.x[0]; row->A = st.x[1]; row->sigma_h = st.x[4]; row->sigma_w = st.x[5]; row->theta = st.x[6];
    row->rmse = fsq_sqrt(rm / 25.0);
    row->r2 = 1.0 - num / den;
    row->s_n = fsq_illumina_s_n(data, vmax);
    row->p2 = st.x[2]; row->p3 = st.x[3];
    row->h = h; row->w = w; row->field = field;
    row->status = status; row->niter = st.niter; row->nfev = st.nfev + (status > 0 ? 1 : 0);
    row->key_h = -1; row->key_w = -1;
}

// Persistent kernel: every lane pulls candidates from a global queue; all lanes of a wave advance one
// outer LM iteration per loop trip, and a lane whose fit has terminated refills immediately, so the
// 1..200 iteration spread of the fits does not leave lanes idle.
template <bool ALIASED, bool FROM_IMAGE>
__global__ void __launch_bounds__(64) fsq_fit_persistent(const uint16_t* __restrict__ img, int H, int W,
                                                         const int32_t* __restrict__ cand, long long n,
                                                         FsqRow* __restrict__ rows, unsigned long long* __restrict__ queue)
{
    FsqLmState st;
    double data[FSQ_NPIX];
    double vmax = 0., vmean = 0.;
    long long idx = -1;
    int h = 2, w = 2, field = 0;
    bool active = false, drained = false;
    for (;;) {
        if (!active && !drained) {
            idx = (long long)atomicAdd(queue, 1ull);
            if (idx < n) {
                if (FROM_IMAGE) {
                    field = cand[3 * idx]; h = cand[3 * idx + 1]; w = cand[3 * idx + 2];
                    const uint16_t* base = img + ((size_t)field * H + (h - 2)) * W + (w - 2);
                    for (int a = 0; a < 5; a++)
                        for (int b = 0; b < 5; b++) data[a * 5 + b] = (double)base[(size_t)a * W + b];
                } else {
                    for (int k = 0; k < FSQ_NPIX; k++) data[k] = (double)img[idx * FSQ_NPIX + k];
                }
                double vmedian;
                fsq_roi_stats(data, &vmedian, &vmax, &vmean);
                fsq_lm_init(data, vmedian, vmax, vmean, st);
                active = true;
            } else {
                drained = true;
            }
        }
        if (!__any(active)) break;
        if (active) {
            int status = fsq_lm_outer<ALIASED>(data, st);
            if (status != 0) {
                fsq_finish_row(data, st, status, vmax, vmean, h, w, field, &rows[idx]);
                active = false;
            }
        }
    }
}

#endif  // FSQ_BUILD_AB

__global__ void fsq_fit_images_kernel(const FsqRow* __restrict__ rows, const int32_t* __restrict__ idx, int64_t n,
                                      double* __restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FsqRow& r = rows[idx ? idx[i] : i];
    double p[FSQ_NP] = {r.H, r.A, r.p2, r.p3, r.sigma_h, r.sigma_w, r.theta};
    double g[FSQ_NPIX];
    fsq_model(p, g);
    for (int k = 0; k < FSQ_NPIX; k++) out[i * FSQ_NPIX + k] = g[k];
}

#ifdef FSQ_BUILD_AB
static int fsq_launch_fit(const uint16_t* d_src, int H, int W, const int32_t* d_cand, int64_t n, int mode, bool from_image,
                          FsqRow* d_rows, hipStream_t s)
{
    static std::atomic<unsigned> next_slot{0};
    unsigned slot = next_slot.fetch_add(1) % 256u;
    unsigned long long* queue = nullptr;
    FSQ_HIP_CHECK(hipGetSymbolAddress((void**)&queue, HIP_SYMBOL(g_fit_queue)));
    queue += slot;
    FSQ_HIP_CHECK(hipMemsetAsync(queue, 0, sizeof(unsigned long long), s));
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long waves = (n + 63) / 64;
    long long resident = (long long)cus * 8;               // persistent waves; more than fit at once is harmless
    dim3 grid((unsigned)(waves < resident ? waves : resident)), block(64);
    const bool ref = (mode == FSQ_MODE_REF);
    if (from_image) {
        if (ref) hipLaunchKernelGGL((fsq_fit_persistent<true, true>), grid, block, 0, s, d_src, H, W, d_cand, (long long)n, d_rows, queue);
        else hipLaunchKernelGGL((fsq_fit_persistent<false, true>), grid, block, 0, s, d_src, H, W, d_cand, (long long)n, d_rows, queue);
    } else {
        if (ref) hipLaunchKernelGGL((fsq_fit_persistent<true, false>), grid, block, 0, s, d_src, 5, 5, d_cand, (long long)n, d_rows, queue);
        else hipLaunchKernelGGL((fsq_fit_persistent<false, false>), grid, block, 0, s, d_src, 5, 5, d_cand, (long long)n, d_rows, queue);
    }
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}

int fsq_launch_fit_quad(const uint16_t* d_src, int H, int W, const int32_t* d_cand, int64_t n, int mode, bool from_image,
                        FsqRow* d_rows, void* d_ws, int64_t ws_bytes, hipStream_t s);
#endif
int fsq_launch_fit_rounds(const uint16_t* d_src, int H, int W, const int32_t* d_cand, int64_t n, int mode, bool from_image,
                          FsqRow* d_rows, void* d_ws, int64_t ws_bytes, hipStream_t s);

extern "C" int fsq_fit_candidates(const uint16_t* d_img, int n_fields, int H, int W, const int32_t* d_cand, int64_t n,
                                  int mode, FsqRow* d_rows, void* d_workspace, int64_t workspace_bytes, void* stream)
{
    const int m = mode & 0xff;
    if (n < 0 || H < 5 || W < 5 || n_fields < 1 || (m != FSQ_MODE_REF && m != FSQ_MODE_TEXTBOOK && m != FSQ_MODE_TEXTBOOK_F32)) return FSQ_EINVAL;
    if (m == FSQ_MODE_TEXTBOOK_F32 && (mode & (FSQ_ENGINE_LANE | FSQ_ENGINE_QUAD))) return FSQ_EINVAL;
    if (n == 0) return FSQ_OK;
    if (!d_img || !d_cand || !d_rows) return FSQ_EINVAL;
#ifndef FSQ_BUILD_AB
    if (mode & (FSQ_ENGINE_LANE | FSQ_ENGINE_QUAD)) return FSQ_ENOTIMPL;          // the A/B engines are not in this build
#else
    if ((mode & FSQ_PIXELS_F16_FLAG) && (mode & (FSQ_ENGINE_LANE | FSQ_ENGINE_QUAD))) return FSQ_ENOTIMPL;   // A/B engines: uint16 only
#endif
    if (mode & FSQ_PIXELS_U32_FLAG) {       // uint32 pixels (round 4): the rounds engine's 32-bit instantiations, fp64 modes only
        if ((mode & (FSQ_PIXELS_F16_FLAG | FSQ_ENGINE_LANE | FSQ_ENGINE_QUAD)) || m == FSQ_MODE_TEXTBOOK_F32) return FSQ_ENOTIMPL;
        return fsq_launch_fit_rounds(d_img, H, W, d_cand, n, m | FSQ_PIXELS_U32_FLAG, true, d_rows, d_workspace, workspace_bytes, (hipStream_t)stream);
    }
    if (mode & FSQ_PIXELS_F16_FLAG)
        return fsq_launch_fit_rounds(d_img, H, W, d_cand, n, m | FSQ_PIXELS_F16_FLAG, true, d_rows, d_workspace, workspace_bytes, (hipStream_t)stream);
#ifdef FSQ_BUILD_AB
    if (mode & FSQ_ENGINE_LANE) return fsq_launch_fit(d_img, H, W, d_cand, n, m, true, d_rows, (hipStream_t)stream);
    if (mode & FSQ_ENGINE_QUAD) return fsq_launch_fit_quad(d_img, H, W, d_cand, n, m, true, d_rows, d_workspace, workspace_bytes, (hipStream_t)stream);
#endif
    return fsq_launch_fit_rounds(d_img, H, W, d_cand, n, m, true, d_rows, d_workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int fsq_fit_rois(const uint16_t* d_rois, int64_t n, int mode, FsqRow* d_rows, void* d_workspace,
                            int64_t workspace_bytes, void* stream)
{
    const int m = mode & 0xff;
    if (n < 0 || (m != FSQ_MODE_REF && m != FSQ_MODE_TEXTBOOK && m != FSQ_MODE_TEXTBOOK_F32)) return FSQ_EINVAL;
    if (m == FSQ_MODE_TEXTBOOK_F32 && (mode & (FSQ_ENGINE_LANE | FSQ_ENGINE_QUAD))) return FSQ_EINVAL;
    if (n == 0) return FSQ_OK;
    if (!d_rois || !d_rows) return FSQ_EINVAL;
#ifdef FSQ_BUILD_AB
    if (mode & FSQ_ENGINE_LANE) return fsq_launch_fit(d_rois, 5, 5, nullptr, n, m, false, d_rows, (hipStream_t)stream);
    if (mode & FSQ_ENGINE_QUAD) return fsq_launch_fit_quad(d_rois, 5, 5, nullptr, n, m, false, d_rows, d_workspace, workspace_bytes, (hipStream_t)stream);
#else
    if (mode & (FSQ_ENGINE_LANE | FSQ_ENGINE_QUAD)) return FSQ_ENOTIMPL;
#endif
    return fsq_launch_fit_rounds(d_rois, 5, 5, nullptr, n, m, false, d_rows, d_workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int fsq_fit_images(const FsqRow* d_rows, const int32_t* d_idx, int64_t n, double* d_fit_img, void* stream)
{
    if (n < 0) return FSQ_EINVAL;
    if (n == 0) return FSQ_OK;
    if (!d_rows || !d_fit_img) return FSQ_EINVAL;
    hipLaunchKernelGGL(fsq_fit_images_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, d_rows, d_idx, n, d_fit_img);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}

extern "C" int fsq_has_ab_engines(void)
{
#ifdef FSQ_BUILD_AB
    return 1;
#else
    return 0;
#endif
}
