// Microbenchmark (gfx950): do fp64 MFMA (v_mfma_f64_16x16x4_f64) waves and fp64 VALU FMA waves on the SAME SIMD overlap?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) run VALU FMA chains, waves 4-7 run MFMA chains.
// mode 0: only the VALU waves work, 1: only the MFMA waves, 2: both.  Prints wall time and the equivalent FMA rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define ITERS 4000
__global__ __launch_bounds__(512) void k(double* out, int mode, int valu_n, int mfma_n) {
    const int wave = threadIdx.x >> 6;
    double acc = 0.0;
    if (wave < 4) {
        if (mode == 1) return;
        double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; const double b = 1.0000001, c = 1e-9;
        for (int i = 0; i < valu_n; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            }
        }
        acc = a0 + a1 + a2 + a3;
    } else {
        if (mode == 0) return;
        double4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
        for (int i = 0; i < mfma_n; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            }
        }
        acc = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    double* out; hipMalloc(&out, sizeof(double) * 256 * 512);
    const int valu_n = ITERS, mfma_n = ITERS / 4;     // per wave: valu_n*16 FMA instr (64 lanes each); mfma_n*4 MFMA instr (1024 FMA-equiv each)
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, valu_n, mfma_n);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, valu_n, mfma_n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double valu_fma = (mode != 1) ? 256.0 * 4 * valu_n * 16 * 64 : 0, mfma_fma = (mode != 0) ? 256.0 * 4 * mfma_n * 4 * 1024 : 0;
        printf("mode %d: %.3f ms  VALU %.1f TFLOP/s  MFMA %.1f TFLOP/s (fp64, FMA = 2 flop)\n", mode, ms, 2 * valu_fma / ms / 1e9, 2 * mfma_fma / ms / 1e9);
    }
    return 0;
}
