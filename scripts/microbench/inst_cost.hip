// Microbenchmark: issue cost / dependent latency of the fp64 instructions the Voigt kernel uses.
// One block per CU-ish, W waves per block; each wave runs a timed loop and reports cycles/instr.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define N_ITERS 2000
#define REP8(x) x x x x x x x x

template <int KIND>
__global__ void bench(double* out, unsigned long long* cyc, double seed) {
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
    double a1 = a + 1, a2 = a + 2, a3 = a + 3;
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    int iacc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < N_ITERS; ++i) {
        if (KIND == 0) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (KIND == 1) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (KIND == 2) { REP8(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 3) { REP8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));) }
        if (KIND == 4) { REP8(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) }
        if (KIND == 5) { REP8(asm volatile("v_rcp_f64 %0, %4\n v_rcp_f64 %1, %4\n v_rcp_f64 %2, %4\n v_rcp_f64 %3, %4" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a));) }
        if (KIND == 6) { REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");) }
        if (KIND == 7) { REP8(asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(iacc) : "s20");) }
        if (KIND == 8) { REP8(asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 9) { REP8(a = exp(-a * 1e-3) + 1.0;) }
        if (KIND == 10) { REP8(asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(iacc));) }
        if (KIND == 11) { REP8(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(iacc) : "v"(a));) }
        if (KIND == 12) { float f = (float)a; REP8(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));) a = f; }
        if (KIND == 13) { REP8(asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(r0) : "v"(a), "v"(b));) }   // independent (no RAW)
        if (KIND == 14) { REP8(asm volatile("v_rndne_f64 %0, %0" : "+v"(a));) }
        if (KIND == 15) { REP8(asm volatile("v_mov_b32 %0, %1" : "=v"(iacc) : "v"(iacc));) }
        if (KIND == 16) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iacc) : "v"(iacc) : "vcc");) }
        if (KIND == 17) { REP8(asm volatile("v_fma_f64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(b) : "s20", "s21");) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a1 + a2 + a3 + r0 + r1 + r2 + r3 + iacc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter) {
    for (int waves : {4, 8, 16, 32}) {   // waves per CU (block of waves*64 threads... max 1024 threads)
        int threads = waves * 64;
        int blocks = 256;
        if (threads > 1024) { blocks = 256 * (threads / 1024); threads = 1024; }
        double* out; unsigned long long* cyc;
        hipMalloc(&out, sizeof(double) * blocks * threads);
        hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 16);
        hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.5);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.5);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * (threads / 64));
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= h.size();
        double n_inst = (double)N_ITERS * 8 * per_iter;
        // s_memtime ticks at 100 MHz on gfx9? report both raw ticks/inst and wall-derived cycles at 2.4 GHz
        double wall_cyc_per_inst_per_simd = (ms * 1e-3 * 2.4e9) / (n_inst * waves / 4.0);
        printf("%-28s waves/CU=%2d  memtime/inst=%7.3f  wall: %6.2f cyc per inst per SIMD (at 2.4GHz), %7.3f ms\n", name, waves,
               avg / n_inst, wall_cyc_per_inst_per_simd, ms);
        hipFree(out); hipFree(cyc);
    }
}

int main() {
    run<0>("fma_f64 dependent", 1);
    run<1>("fma_f64 4 indep chains", 4);
    run<13>("fma_f64 no RAW", 1);
    run<17>("fma_f64 dep, SGPR operand", 1);
    run<2>("mul_f64 dependent", 1);
    run<3>("add_f64 dependent", 1);
    run<4>("rcp_f64 dependent", 1);
    run<5>("rcp_f64 independent x4", 4);
    run<6>("cmp_lt_f64", 1);
    run<7>("readlane_b32", 1);
    run<8>("min_f64 dependent", 1);
    run<9>("exp(double) ocml + add", 1);
    run<10>("ldexp_f64 dep", 1);
    run<11>("cvt_i32_f64", 1);
    run<14>("rndne_f64 dep", 1);
    run<12>("fma_f32 dependent", 1);
    run<15>("v_mov_b32", 1);
    run<16>("v_cndmask_b32", 1);
    return 0;
}
