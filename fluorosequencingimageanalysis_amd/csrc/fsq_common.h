// fsq_common.h - shared bits of the HIP translation units
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fsq.h"

extern thread_local hipError_t g_fsq_last_hip;
#define FSQ_HIP_CHECK(expr)                          \
    do {                                             \
        hipError_t e_ = (expr);                      \
        if (e_ != hipSuccess) { g_fsq_last_hip = e_; return FSQ_EHIP; } \
    } while (0)

// Pixel storage formats of the image arguments (include/fsq.h).  FSQ_PIXELS_F16: IEEE binary16 values holding the
// (pre-scaled) intensities; a pixel's integer value is the half truncated toward zero - what the reference's
// image.astype(np.int64) (pflib.py:241, 443) gives for a float16 image.  Negative / NaN -> 0, +inf -> 65535.
__device__ __forceinline__ unsigned fsq_pixel(const uint16_t* p, size_t i, int fmt)
{
    if (fmt == 2) return ((const uint32_t*)p)[i];    // FSQ_PIXELS_U32 (the pointer is the image's base: i counts pixels)
    const unsigned raw = p[i];
    if (fmt == 0) return raw;
    const unsigned e = (raw >> 10) & 31u, m = raw & 1023u;
    if (raw & 0x8000u) return 0u;
    if (e == 31u) return m ? 0u : 65535u;
    if (e < 15u) return 0u;                          // |v| < 1 (incl. subnormals)
    const unsigned v = (1024u | m);                  // 1.m x 2^(e-15), as an integer shifted by 10
    return e >= 25u ? (v << (e - 25u)) : (v >> (25u - e));
}
