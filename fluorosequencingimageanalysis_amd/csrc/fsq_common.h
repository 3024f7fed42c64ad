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
