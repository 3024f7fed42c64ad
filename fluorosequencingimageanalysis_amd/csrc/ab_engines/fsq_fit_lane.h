// fsq_fit_lane.h - the one-lane-per-fit persistent LM engine (FSQ_ENGINE_LANE of include/fsq.h).  NOT part of the shipped
// library: it is compiled only into the A/B build (`make ab`, -DFSQ_BUILD_AB), where it serves as an independent second
// implementation for timing comparisons and for tests/test_gpu_fit.py::test_lane_engine_equals_quad_engine.  Included by
// ../fsq_fit.hip under FSQ_BUILD_AB.
#pragma once
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

