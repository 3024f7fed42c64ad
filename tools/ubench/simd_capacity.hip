// Micro-benchmark: how many fp64 VALU instructions per cycle a SIMD sustains with 1, 2, 4, 8 waves resident, on the whole
// chip (so that clock management under load is included).  hipcc --offload-arch=gfx950 -O3 -o simd_capacity simd_capacity.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void __launch_bounds__(256) k(double* out, int iters)
{
    double a[CHAINS];
    for (int c = 0; c < CHAINS; c++) a[c] = 1.0 + threadIdx.x * 1e-9 + c;
    const double m = 1.0000001, b = 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) a[c] = __builtin_fma(a[c], m, b);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c];
    if (s == 12345.678) out[0] = s;
}
template <int CHAINS>
void run(int waves_per_simd, int cus)
{
    double* out; hipMalloc(&out, 64);
    const int iters = 20000;
    // blocks of 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU resident (LDS-free, few registers)
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS>), grid, block, 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS>), grid, block, 0, 0, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 8 * CHAINS;
    const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("chains=%d waves/simd=%d cus=%d: %.3f ms, %.3f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz), %.1f TFLOP/s\n", CHAINS,
           waves_per_simd, cus, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4,
           instr_per_wave * waves_per_simd * 4.0 * cus * 128.0 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main()
{
    int dev = 0, cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    for (int w : {1, 2, 4, 8}) { run<1>(w, cus); run<4>(w, cus); }
    run<4>(2, 1); run<4>(4, 1);
    return 0;
}
