// fsq_fit.hip - K3/K4: per-candidate LM PSF fit + fit-quality metrics, one lane per candidate.
// Reference: the candidate loop of pflib.find_peptides, pflib.py:441-477.
#include <atomic>

#include "fsq_common.h"
#include "fsq_lm_core.h"

#ifdef FSQ_BUILD_AB      // the A/B engines (make ab): not in the shipped library
#include "ab_engines/fsq_fit_lane.h"
#endif

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
