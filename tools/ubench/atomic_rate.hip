// Micro-benchmark: chip-wide throughput of wave-aggregated queue-tail atomics (one atomicAdd WITH return per wave), the
// pattern of wave_reserve in csrc/fsq_fit_rounds.hip.  Variants: all waves on ONE counter; on two counters in the same
// 128-byte line (the B-lo / B-hi tails of one counter set); on K counters in K different lines (blockIdx % K).
// Also: the same with the result not consumed (no wait), and with ~20 us of independent fp64 work between issue and use.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) k(int* counters, int stride_ints, int K, int per_wave, int work_iters, int consume, double* out)
{
    const int lane = threadIdx.x;
    int acc = 0;
    double a = 1.0 + lane * 1e-9;
    for (int r = 0; r < per_wave; r++) {
        int* c = counters + (size_t)((blockIdx.x + r) % K) * stride_ints;
        int base = 0;
        if (lane == 0) base = atomicAdd(c, 1);
        for (int i = 0; i < work_iters; i++) a = __builtin_fma(a, 1.0000001, 1e-9);
        if (consume) acc += __shfl(base, 0);
    }
    if (acc == 0x7fffffff || a == 12345.678) out[0] = acc + a;
}
static void run(const char* name, int waves, int K, int stride_ints, int per_wave, int work_iters, int consume)
{
    int* counters; double* out;
    hipMalloc(&counters, (size_t)K * stride_ints * 4 + 4096); hipMalloc(&out, 64);
    hipMemset(counters, 0, (size_t)K * stride_ints * 4 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, counters, stride_ints, K, per_wave, work_iters, consume, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, counters, stride_ints, K, per_wave, work_iters, consume, out);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves=%7d x%d K=%3d work=%5d consume=%d: %8.3f ms  %8.1f atomics/us\n", name, waves, per_wave, K, work_iters, consume, ms,
           (double)waves * per_wave / (ms * 1e3));
    hipFree(counters); hipFree(out);
}
int main()
{
    const int W = 200000;
    run("one counter", W, 1, 32, 1, 0, 1);
    run("one counter, 2 per wave", W, 1, 32, 2, 0, 1);
    run("two counters in one line", W, 2, 1, 1, 0, 1);
    run("2 lines", W, 2, 32, 1, 0, 1);
    run("4 lines", W, 4, 32, 1, 0, 1);
    run("8 lines", W, 8, 32, 1, 0, 1);
    run("16 lines", W, 16, 32, 1, 0, 1);
    run("64 lines", W, 64, 32, 1, 0, 1);
    run("16 lines 4 KB apart", W, 16, 1024, 1, 0, 1);
    run("one counter, result not consumed", W, 1, 32, 1, 0, 0);
    run("one counter, 2000 fma between", W, 1, 32, 1, 2000, 1);
    run("no atomics: 2000 fma only (K irrelevant)", W, 1, 32, 0, 2000, 1);
    run("16 lines, 2000 fma between", W, 16, 32, 1, 2000, 1);
    run("one counter, 8000 fma between", W / 4, 1, 32, 1, 8000, 1);
    run("16 lines, 8000 fma between", W / 4, 16, 32, 1, 8000, 1);
    return 0;
}
