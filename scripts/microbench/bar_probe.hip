// Microbenchmark (gfx950): how fast does a waiting kernel learn that the host has set a flag?
//  A: flag in pinned host memory, polled by the kernel over PCIe (system-scope loads)
//  B: flag in fine-grained DEVICE memory written by the CPU through the PCIe BAR (if the allocation is CPU-accessible at all),
//     polled by the kernel locally
// The kernel acknowledges into pinned host memory; the host measures flag store -> ack seen (round trip), median of many.
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
#include <algorithm>
static sigjmp_buf g_jb;
static void on_segv(int) { siglongjmp(g_jb, 1); }
__global__ void waiter(const unsigned int* flag, unsigned int* ack, unsigned int want, int n_pollers) {
    // every workgroup polls (n_pollers = gridDim.x) or only workgroup 0
    if ((int)blockIdx.x >= n_pollers) return;
    const long long t0 = wall_clock64();
    unsigned int v = 0;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != want) {
        if (wall_clock64() - t0 > 100000000ll) break;     // 1 s
        __builtin_amdgcn_s_sleep(1);
    }
    if (threadIdx.x == 0) __hip_atomic_store(ack + 16 * blockIdx.x, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double trial(const unsigned int* d_flag, volatile unsigned int* h_flag_cpu, unsigned int* d_ack, volatile unsigned int* h_ack, int grid, int n_pollers, int reps, const char* what) {
    std::vector<double> us;
    for (int r = 1; r <= reps; ++r) {
        for (int b = 0; b < grid; ++b) h_ack[16 * b] = 0;
        hipLaunchKernelGGL(waiter, dim3(grid), dim3(64), 0, 0, d_flag, d_ack, (unsigned int)r, n_pollers);
        auto t = std::chrono::steady_clock::now();
        while (std::chrono::steady_clock::now() - t < std::chrono::microseconds(60)) { }     // the kernel is up and polling
        auto t0 = std::chrono::steady_clock::now();
        *h_flag_cpu = (unsigned int)r;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        int last = n_pollers - 1;
        while (h_ack[16 * last] != (unsigned int)r || h_ack[0] != (unsigned int)r) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { printf("%s: timeout\n", what); return -1; }
        }
        // all pollers
        for (int b = 0; b < n_pollers; ++b) while (h_ack[16 * b] != (unsigned int)r) { }
        auto t1 = std::chrono::steady_clock::now();
        us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        hipDeviceSynchronize();
    }
    std::sort(us.begin(), us.end());
    printf("%s grid %d pollers %d: flag store -> all acks seen: median %.2f us, p10 %.2f, p90 %.2f\n", what, grid, n_pollers, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
    return us[us.size() / 2];
}
int main() {
    unsigned int *h_flag, *h_ack, *d_hflag, *d_ack;
    hipHostMalloc((void**)&h_flag, 4096, hipHostMallocMapped); hipHostMalloc((void**)&h_ack, 16 * 4 * 1024, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&d_hflag, h_flag, 0); hipHostGetDevicePointer((void**)&d_ack, h_ack, 0);
    *h_flag = 0;
    for (int np : {1, 8, 64, 512}) trial(d_hflag, h_flag, d_ack, h_ack, 512, np, 200, "A pinned-host flag");
    // B: device memory the CPU can write?
    unsigned int* d_fg = nullptr;
    for (int kind = 0; kind < 3; ++kind) {
        const char* name = kind == 0 ? "B fine-grained device flag" : kind == 1 ? "B uncached device flag" : "B managed (device-preferred) flag";
        hipError_t e;
        if (kind == 0) e = hipExtMallocWithFlags((void**)&d_fg, 4096, hipDeviceMallocFinegrained);
        else if (kind == 1) e = hipExtMallocWithFlags((void**)&d_fg, 4096, hipDeviceMallocUncached);
        else {
            e = hipMallocManaged((void**)&d_fg, 4096);
            if (e == hipSuccess) { hipMemAdvise(d_fg, 4096, hipMemAdviseSetPreferredLocation, 0); hipMemAdvise(d_fg, 4096, hipMemAdviseSetAccessedBy, hipCpuDeviceId); hipMemPrefetchAsync(d_fg, 4096, 0, 0); hipDeviceSynchronize(); }
        }
        if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipMemset(d_fg, 0, 4096); hipDeviceSynchronize();
        struct sigaction sa{}, old{}; sa.sa_handler = on_segv; sigaction(SIGSEGV, &sa, &old); struct sigaction oldb{}; sigaction(SIGBUS, &sa, &oldb);
        bool ok = false;
        if (sigsetjmp(g_jb, 1) == 0) { *(volatile unsigned int*)d_fg = 0u; ok = true; }
        sigaction(SIGSEGV, &old, nullptr); sigaction(SIGBUS, &oldb, nullptr);
        if (!ok) { printf("%s: the CPU cannot write it (fault)\n", name); continue; }
        for (int np : {1, 64, 512}) trial(d_fg, d_fg, d_ack, h_ack, 512, np, 200, name);
        if (kind < 2) {
            // how long does the CPU take to put 24 KB there (a 512 x 6 theta batch)?  (write-combined or one transaction per store?)
            unsigned char* big = nullptr;
            if (hipExtMallocWithFlags((void**)&big, 1 << 16, kind == 0 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached) == hipSuccess) {
                static unsigned char src[24576];
                for (int i = 0; i < 24576; ++i) src[i] = (unsigned char)i;
                std::vector<double> us;
                for (int r = 0; r < 200; ++r) {
                    auto t0 = std::chrono::steady_clock::now();
                    memcpy(big, src, 24576);
                    __builtin_ia32_sfence();
                    auto t1 = std::chrono::steady_clock::now();
                    us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
                }
                std::sort(us.begin(), us.end());
                printf("%s: CPU memcpy of 24 KB into it + sfence: median %.2f us (p90 %.2f)\n", name, us[100], us[180]);
                hipFree(big);
            }
        }
    }
    return 0;
}
