// fsq_capi.hip - library-level entry points of the C ABI (include/fsq.h)
#include "fsq_common.h"

thread_local hipError_t g_fsq_last_hip = hipSuccess;

extern "C" const char* fsq_version(void) { return "fsq-hip 0.4 (gfx950)"; }
extern "C" const char* fsq_last_hip_error(void) { return hipGetErrorString(g_fsq_last_hip); }
extern "C" int fsq_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

