// Microbenchmark: how many single-wave workgroups are resident per CU as a function of dynamic LDS
// bytes and VGPR count (gfx950).  Each wave spins ~30 us; residency = max overlap of [start,end].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int NV>
__global__ __launch_bounds__(64) void spin(unsigned long long* st, int ticks) {
    extern __shared__ double lds[];
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = acc[i] * 1.0000001 + 1e-9;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += acc[i];
    if (s == 12345.678) lds[threadIdx.x] = s;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t0; st[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}

template <int NV>
void run(int lds_bytes) {
    const int nb = 16384;
    unsigned long long* d; hipMalloc(&d, nb * 16);
    hipLaunchKernelGGL(spin<NV>, dim3(nb), dim3(64), lds_bytes, 0, d, 3000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    std::vector<std::pair<unsigned long long, int>> ev;
    for (int i = 0; i < nb; ++i) { ev.push_back({h[2 * i], 1}); ev.push_back({h[2 * i + 1], -1}); }
    std::sort(ev.begin(), ev.end());
    int cur = 0, mx = 0;
    for (auto& e : ev) { cur += e.second; mx = std::max(mx, cur); }
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)spin<NV>);
    printf("accs=%2d numRegs=%3d lds=%6d B -> max resident waves %5d (%.1f per CU, %.2f per SIMD)\n", NV, fa.numRegs, lds_bytes, mx,
           mx / 256.0, mx / 1024.0);
    hipFree(d);
}

int main() {
    for (int l : {0, 2048, 4096, 5120, 6144, 6392, 6656, 7168, 8192, 9216, 10240}) run<30>(l);
    for (int l : {0, 4096, 6392}) run<20>(l);
    for (int l : {0, 4096, 6392}) run<38>(l);
    return 0;
}
