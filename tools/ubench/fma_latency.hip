// Micro-benchmark: issue cost of dependent vs independent fp64 VALU chains on one SIMD, at 1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o fma_latency fma_latency.hip ; ./fma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, int KIND>
__global__ void k(double* out, int iters, long long* cyc)
{
    double a[CHAINS];
    for (int c = 0; c < CHAINS; c++) a[c] = 1.0 + threadIdx.x * 1e-9 + c;
    const double m = 1.0000001, b = 1e-9;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) a[c] = __builtin_fma(a[c], m, b);
                else if (KIND == 1) a[c] = a[c] * m;
                else if (KIND == 2) a[c] = a[c] + b;
                else if (KIND == 3) a[c] = __builtin_amdgcn_rcp(a[c]);
                else if (KIND == 4) a[c] = __builtin_amdgcn_div_fixup(a[c], m, b);
                else if (KIND == 5) a[c] = (double)__builtin_amdgcn_frexp_exp(a[c]) + b;
                else if (KIND == 6) a[c] = __builtin_amdgcn_sqrt(a[c]);
            }
    }
    long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAINS, int KIND>
void run(const char* name, int waves_per_simd)
{
    double* out; long long* cyc; long long h;
    hipMalloc(&out, 8 * 64 * 8 * 1024); hipMalloc(&cyc, 8);
    const int iters = 2000;
    // one CU: blocks of 64*4*waves_per_simd threads -> waves_per_simd waves on each of the 4 SIMDs
    hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(1), dim3(64 * 4 * waves_per_simd), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(1), dim3(64 * 4 * waves_per_simd), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double per = (double)h / ((double)iters * 8 * CHAINS);
    printf("%-10s chains=%d waves/simd=%d : %.2f clk64-ticks per instr per wave  (x%d waves)\n", name, CHAINS, waves_per_simd, per, waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main()
{
    // s_memtime/clock64 counts at a fixed 100 MHz on gfx9; calibrate with wall time instead
    for (int w = 1; w <= 2; w++) {
        run<1, 0>("fma", w); run<2, 0>("fma", w); run<4, 0>("fma", w); run<8, 0>("fma", w);
        run<1, 1>("mul", w); run<4, 1>("mul", w);
        run<1, 2>("add", w); run<4, 2>("add", w);
        run<1, 3>("rcp", w); run<4, 3>("rcp", w);
        run<1, 4>("div_fixup", w); run<4, 4>("div_fixup", w);
        run<1, 5>("frexp+cvt+add", w); run<4, 5>("frexp+cvt+add", w);
        run<1, 6>("sqrt", w); run<4, 6>("sqrt", w);
    }
    // wall-clock calibration of the tick
    double* out; long long* cyc; hipMalloc(&out, 8 * 64 * 8 * 1024); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<8, 0>), dim3(1), dim3(256), 0, 0, out, 200000, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("calibration: %lld ticks in %.3f ms -> %.1f MHz tick; 8-chain fma: %.3f ns per instr per wave\n", h, ms, h / ms / 1e3, ms * 1e6 / (200000.0 * 64));
    return 0;
}
