// fsq_fit.hip - K3/K4: per-candidate LM PSF fit + fit-quality metrics, one lane per candidate.
// Reference: the candidate loop of pflib.find_peptides, pflib.py:441-477.
#include "fsq_common.h"
#include "fsq_lm_core.h"

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

template <bool ALIASED>
FSQ_DEV void fsq_fit_one(const double* data, int h, int w, int field, FsqRow* row)
{
    // start values of pflib._fit_2d_gaussian: median, max, mean of the ROI (pflib.py:201-209)
    double srt[FSQ_NPIX];
    double vmax = data[0], isum = 0.0;
    for (int i = 0; i < FSQ_NPIX; i++) { srt[i] = data[i]; vmax = data[i] > vmax ? data[i] : vmax; isum += data[i]; }
    for (int i = 1; i < FSQ_NPIX; i++) {     // insertion sort (25 values)
        double v = srt[i];
        int j = i - 1;
        while (j >= 0 && srt[j] > v) { srt[j + 1] = srt[j]; j--; }
        srt[j + 1] = v;
    }
    double vmean = isum / 25.0;
    FsqLmResult res;
    fsq_lm_fit<ALIASED>(data, srt[12], vmax, vmean, &res);
    // metrics, pflib.py:461-473
    double fit[FSQ_NPIX];
    fsq_model(res.p, fit);
    double num = 0.0, den = 0.0, rm = 0.0;
    for (int i = 0; i < FSQ_NPIX; i++) { double d = data[i] - fit[i]; num += d * d; }
    for (int i = 0; i < FSQ_NPIX; i++) { double d = data[i] - vmean; den += d * d; }
    for (int i = 0; i < FSQ_NPIX; i++) rm += fsq_pow2(data[i] - fit[i]);
    row->h0 = res.p[2] + h - 2.5;
    row->w0 = res.p[3] + w - 2.5;
    row->H = res.p[0]; row->A = res.p[1]; row->sigma_h = res.p[4]; row->sigma_w = res.p[5]; row->theta = res.p[6];
    row->rmse = fsq_sqrt(rm / 25.0);
    row->r2 = 1.0 - num / den;
    row->s_n = fsq_illumina_s_n(data, vmax);
    row->p2 = res.p[2]; row->p3 = res.p[3];
    row->h = h; row->w = w; row->field = field;
    row->status = res.status; row->niter = res.niter; row->nfev = res.nfev;
    row->key_h = -1; row->key_w = -1;
}

template <bool ALIASED>
__global__ void __launch_bounds__(64) fsq_fit_cand_kernel(const uint16_t* __restrict__ img, int H, int W,
                                                          const int32_t* __restrict__ cand, int64_t n,
                                                          FsqRow* __restrict__ rows)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int field = cand[3 * i], h = cand[3 * i + 1], w = cand[3 * i + 2];
    const uint16_t* base = img + ((size_t)field * H + (h - 2)) * W + (w - 2);
    double data[FSQ_NPIX];
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++) data[a * 5 + b] = (double)base[(size_t)a * W + b];
    fsq_fit_one<ALIASED>(data, h, w, field, &rows[i]);
}

template <bool ALIASED>
__global__ void __launch_bounds__(64) fsq_fit_roi_kernel(const uint16_t* __restrict__ rois, int64_t n,
                                                         FsqRow* __restrict__ rows)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double data[FSQ_NPIX];
    for (int k = 0; k < FSQ_NPIX; k++) data[k] = (double)rois[i * FSQ_NPIX + k];
    fsq_fit_one<ALIASED>(data, 2, 2, 0, &rows[i]);
}

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

extern "C" int fsq_fit_candidates(const uint16_t* d_img, int n_fields, int H, int W, const int32_t* d_cand, int64_t n,
                                  int mode, FsqRow* d_rows, void* stream)
{
    if (n < 0 || H < 5 || W < 5 || n_fields < 1 || (mode != FSQ_MODE_REF && mode != FSQ_MODE_TEXTBOOK)) return FSQ_EINVAL;
    if (n == 0) return FSQ_OK;
    if (!d_img || !d_cand || !d_rows) return FSQ_EINVAL;
    dim3 grid((unsigned)((n + 63) / 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    if (mode == FSQ_MODE_REF) hipLaunchKernelGGL(fsq_fit_cand_kernel<true>, grid, block, 0, s, d_img, H, W, d_cand, n, d_rows);
    else hipLaunchKernelGGL(fsq_fit_cand_kernel<false>, grid, block, 0, s, d_img, H, W, d_cand, n, d_rows);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
}

extern "C" int fsq_fit_rois(const uint16_t* d_rois, int64_t n, int mode, FsqRow* d_rows, void* stream)
{
    if (n < 0 || (mode != FSQ_MODE_REF && mode != FSQ_MODE_TEXTBOOK)) return FSQ_EINVAL;
    if (n == 0) return FSQ_OK;
    if (!d_rois || !d_rows) return FSQ_EINVAL;
    dim3 grid((unsigned)((n + 63) / 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    if (mode == FSQ_MODE_REF) hipLaunchKernelGGL(fsq_fit_roi_kernel<true>, grid, block, 0, s, d_rois, n, d_rows);
    else hipLaunchKernelGGL(fsq_fit_roi_kernel<false>, grid, block, 0, s, d_rois, n, d_rows);
    FSQ_HIP_CHECK(hipGetLastError());
    return FSQ_OK;
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
