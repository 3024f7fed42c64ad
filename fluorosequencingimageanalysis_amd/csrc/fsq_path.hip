// fsq_path.hip - the whole per-field path as ONE entry point of the C ABI: fsq_find_peptides.
// Reference: pflib.find_peptides (pflib.py:284-520) for every field of a batch - candidate detection (:217-258), the LM fit
// of every candidate's 5x5 ROI (:441-475), the R^2 filter, consolidation and re-keying (:466, 477-519) - returning what the
// reference's dict holds per kept peak as one flat record table: the FsqRow (tuple entries 0-6, 9-11, the key), fit_img
// (gaussfitter.py:253) and the 25 pixel words of sub_img (pflib.py:443).
// It composes the stage entry points of this library (fsq_detect, fsq_fit_candidates, fsq_consolidate, fsq_kept_rows,
// fsq_fit_images) on the caller's stream and workspace, so a host program - the reference's Python through ctypes, or
// anything else - needs one call per batch of images and no interpreter between the stages.
#include "fsq_common.h"

namespace {

constexpr int REC_WORDS = FSQ_PEAK_RECORD_BYTES / 2;      // a record is 189 16-bit words: 64 (row) + 100 (fit_img) + 25 (sub_img)
constexpr int REC_WORDS_U32 = FSQ_PEAK_RECORD_BYTES_U32 / 2;   // FSQ_PIXELS_U32: 214 words, sub_img as 25 uint32

size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

struct PathLayout {
    size_t cand, counts, offsets, thr, rows, keep, table, fit, stage, fitws, total;
};

PathLayout path_layout(int n_fields, int H, int W, int64_t cand_cap, int64_t record_cap)
{
    PathLayout L;
    size_t o = 0;
    L.cand = o; o += up256((size_t)cand_cap * 3 * sizeof(int32_t));
    L.counts = o; o += up256(((size_t)n_fields + 1) * sizeof(int32_t));
    L.offsets = o; o += up256(((size_t)n_fields + 1) * sizeof(int32_t));
    L.thr = o; o += up256((size_t)n_fields * sizeof(double));
    L.rows = o; o += up256((size_t)cand_cap * sizeof(FsqRow));
    L.keep = o; o += up256((size_t)cand_cap * sizeof(int32_t));
    L.table = o; o += up256((size_t)record_cap * sizeof(FsqRow));
    L.fit = o; o += up256((size_t)record_cap * 25 * sizeof(double));
    const int64_t a = fsq_detect_workspace_bytes(n_fields, H, W), b = fsq_consolidate_workspace_bytes(n_fields, H, W);
    L.stage = o; o += up256((size_t)(a > b ? a : b));
    L.fitws = o; o += up256((size_t)fsq_fit_workspace_bytes(cand_cap));
    L.total = o;
    return L;
}

// record r = row (64 words) | fit_img (100 words) | the ROI's 25 pixel words as they sit in the image
// (WIDE: the image holds uint32 pixels, two record words each)
template <bool WIDE>
__global__ void __launch_bounds__(256) k_pack_records(const FsqRow* __restrict__ table, const double* __restrict__ fit,
                                                      const uint16_t* __restrict__ img, int H, int W, long long k,
                                                      uint16_t* __restrict__ out)
{
    constexpr int RW = WIDE ? REC_WORDS_U32 : REC_WORDS;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * RW) return;
    const long long r = t / RW;
    const int wd = (int)(t - r * RW);
    uint16_t v;
    if (wd < 64) v = ((const uint16_t*)(table + r))[wd];
    else if (wd < 164) v = ((const uint16_t*)(fit + r * 25))[wd - 64];
    else {
        const int q = wd - 164, p = WIDE ? q >> 1 : q, a = p / 5, b = p - 5 * a;
        const FsqRow& R = table[r];
        const size_t at = ((size_t)R.field * H + (R.h - 2 + a)) * W + (R.w - 2 + b);
        v = WIDE ? img[2 * at + (q & 1)] : img[at];
    }
    out[t] = v;
}

}  // namespace

extern "C" int64_t fsq_find_peptides_workspace_bytes(int n_fields, int H, int W, int64_t cand_cap, int64_t record_cap)
{
    if (n_fields < 1 || H < 5 || W < 5 || cand_cap < 1 || record_cap < 1) return FSQ_EINVAL;
    if (fsq_detect_workspace_bytes(n_fields, H, W) < 0 || fsq_fit_workspace_bytes(cand_cap) < 0) return FSQ_EINVAL;
    return (int64_t)path_layout(n_fields, H, W, cand_cap, record_cap).total;
}

extern "C" int fsq_find_peptides(const void* d_img, int n_fields, int H, int W, const FsqDetectParams* prm, double r2_threshold,
                                 int radius, int py2_round, int mode, int64_t cand_cap, void* d_records, int64_t record_cap,
                                 int32_t* d_record_offsets, int32_t* d_nkeep, int64_t* n_candidates, int64_t* n_records,
                                 void* d_workspace, int64_t workspace_bytes, void* stream)
{
    if (!d_img || !prm || !d_records || !d_record_offsets || !d_nkeep || !d_workspace) return FSQ_EINVAL;
    if (n_fields < 1 || H < 5 || W < 5 || cand_cap < 1 || record_cap < 1 || radius < 2) return FSQ_EINVAL;
    const int64_t need = fsq_find_peptides_workspace_bytes(n_fields, H, W, cand_cap, record_cap);
    if (need < 0) return FSQ_EINVAL;
    if (workspace_bytes < need) return FSQ_ENOMEM;
    const PathLayout L = path_layout(n_fields, H, W, cand_cap, record_cap);
    unsigned char* ws = (unsigned char*)d_workspace;
    hipStream_t s = (hipStream_t)stream;
    int32_t* cand = (int32_t*)(ws + L.cand);
    int32_t* counts = (int32_t*)(ws + L.counts);
    int32_t* offsets = (int32_t*)(ws + L.offsets);
    FsqRow* rows = (FsqRow*)(ws + L.rows);
    int32_t* keep = (int32_t*)(ws + L.keep);
    FsqRow* table = (FsqRow*)(ws + L.table);
    double* fit = (double*)(ws + L.fit);
    if (n_candidates) *n_candidates = 0;
    if (n_records) *n_records = 0;
    int rc = fsq_detect((const uint16_t*)d_img, n_fields, H, W, prm, cand, cand_cap, counts, offsets, (double*)(ws + L.thr),
                        ws + L.stage, (int64_t)(L.fitws - L.stage), stream);
    if (rc != FSQ_OK) return rc;
    int32_t total = 0;
    FSQ_HIP_CHECK(hipMemcpyAsync(&total, counts + n_fields, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    FSQ_HIP_CHECK(hipStreamSynchronize(s));
    if (total < 0) return FSQ_ENOTIMPL;                 // a response image sums to >= 2^53 (see fsq_detect)
    if (n_candidates) *n_candidates = total;
    if (total > cand_cap) return FSQ_ERANGE;            // (counts are complete: call again with cand_cap >= *n_candidates)
    const bool wide = prm->pixel_format == FSQ_PIXELS_U32;
    const int fmode = (mode & 0xff) | (prm->pixel_format == FSQ_PIXELS_F16 ? FSQ_PIXELS_F16_FLAG : 0) | (wide ? FSQ_PIXELS_U32_FLAG : 0);
    rc = fsq_fit_candidates((const uint16_t*)d_img, n_fields, H, W, cand, total, fmode, rows, ws + L.fitws, (int64_t)(L.total - L.fitws), stream);
    if (rc != FSQ_OK) return rc;
    rc = fsq_consolidate(rows, counts, offsets, n_fields, H, W, r2_threshold, radius, py2_round, keep, d_nkeep, ws + L.stage,
                         (int64_t)(L.fitws - L.stage), stream);
    if (rc != FSQ_OK) return rc;
    int32_t kept = 0;
    FSQ_HIP_CHECK(hipMemcpyAsync(&kept, d_nkeep + n_fields, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    FSQ_HIP_CHECK(hipStreamSynchronize(s));
    if (kept < 0) kept = 0;
    if (n_records) *n_records = kept;
    if (kept > record_cap) return FSQ_ERANGE;           // (call again with record_cap >= *n_records)
    rc = fsq_kept_rows(rows, keep, offsets, d_nkeep, n_fields, table, record_cap, d_record_offsets, stream);
    if (rc != FSQ_OK) return rc;
    if (kept > 0) {
        rc = fsq_fit_images(table, nullptr, kept, fit, stream);
        if (rc != FSQ_OK) return rc;
        const long long words = (long long)kept * (wide ? REC_WORDS_U32 : REC_WORDS);
        if (wide) hipLaunchKernelGGL(k_pack_records<true>, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, table, fit, (const uint16_t*)d_img,
                                     H, W, (long long)kept, (uint16_t*)d_records);
        else hipLaunchKernelGGL(k_pack_records<false>, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, table, fit, (const uint16_t*)d_img,
                                H, W, (long long)kept, (uint16_t*)d_records);
        FSQ_HIP_CHECK(hipGetLastError());
    }
    return FSQ_OK;
}
