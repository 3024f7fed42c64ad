// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD, 4 waves per SIMD resident, independent operands) of the
// non-arithmetic fp64 helpers the fit kernels use next to v_fma_f64: v_frexp_exp_i32_f64, v_ldexp_f64, v_cndmask (64-bit select =
// 2), v_mov_b64 / 2 x v_mov_b32, v_min_i32, v_bfe_u32, v_cmp_f64, v_div_fixup_f64, ds_bpermute.
//   hipcc --offload-arch=gfx950 -O3 -o op_rates op_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
enum { OP_FMA, OP_FREXP, OP_LDEXP, OP_SELECT, OP_MIN32, OP_BFE, OP_CMP, OP_FIXUP, OP_BPERM, OP_MUL, OP_N };
static const char* NAMES[] = {"v_fma_f64", "v_mul_f64 + v_frexp_exp_i32_f64 + v_add_u32", "v_ldexp_f64", "v_mul_f64 + v_cmp_i32 + 64-bit select (2 v_cndmask)", "v_min_i32", "v_bfe_u32",
                              "v_cmp_lt_f64 + v_cndmask_b32", "v_div_fixup_f64", "ds_bpermute_b32", "v_mul_f64"};
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed)
{
    double a[8]; int b[8];
    for (int c = 0; c < 8; c++) { a[c] = seed + threadIdx.x * 1e-9 + c; b[c] = threadIdx.x + c; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if (OP == OP_FMA) a[c] = __builtin_fma(a[c], 1.0000001, 1e-9);
            if (OP == OP_MUL) a[c] = a[c] * 1.0000001;
            if (OP == OP_FREXP) { a[c] = a[c] * 1.0000001; b[c] += __builtin_amdgcn_frexp_exp(a[c]); }      // (= v_mul_f64 + v_frexp_exp + v_add_u32)
            if (OP == OP_LDEXP) a[c] = __builtin_ldexp(a[c], b[c] & 1);
            if (OP == OP_SELECT) { a[c] = a[c] * 1.0000001; a[c] = (b[c] > i) ? a[c] : seed; }               // (= v_mul_f64 + v_cmp_gt_i32 + 2 v_cndmask)
            if (OP == OP_MIN32) b[c] = min(b[c], b[(c + 1) & 7] + i);
            if (OP == OP_BFE) b[c] = ((unsigned)(b[c] + i) >> 7) & 0x7ff;
            if (OP == OP_CMP) b[c] = (a[c] < (double)b[(c + 1) & 7]) ? b[c] : i;
            if (OP == OP_FIXUP) a[c] = __builtin_amdgcn_div_fixup(a[c], 1.0000001, a[(c + 1) & 7]);
            if (OP == OP_BPERM) b[c] = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 1) << 2, b[c]);
        }
    }
    double s = 0;
    for (int c = 0; c < 8; c++) s += a[c] + b[c];
    if (s == 12345.678) out[0] = s;
}
template <int OP>
void run(int cus)
{
    double* out; (void)hipMalloc(&out, 64);
    const int iters = 4000, waves_per_simd = 4;
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, out, 10, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), grid, block, 0, 0, out, iters, 1.0);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8 * waves_per_simd;          // loop bodies per SIMD
    printf("%-48s %7.3f ms  %6.2f cycles per loop body per SIMD at 2.4 GHz\n", NAMES[OP], ms, ms * 1e-3 * 2.4e9 / n);
    (void)hipFree(out);
}
int main()
{
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    run<OP_FMA>(cus); run<OP_MUL>(cus); run<OP_FREXP>(cus); run<OP_LDEXP>(cus); run<OP_SELECT>(cus); run<OP_MIN32>(cus); run<OP_BFE>(cus); run<OP_CMP>(cus);
    run<OP_FIXUP>(cus); run<OP_BPERM>(cus);
    return 0;
}
