// GPU box: what a pass costs when a RESIDENT kernel is told to go by a stream memory operation instead of being launched:
//   stream S:  hipStreamWriteValue64(go = k)   hipStreamWaitValue64(done >= k)        (no kernel launch on S at all)
//   stream I:  one launch of `resident`: every workgroup polls `go`, "works" for a few hundred cycles, the last one to
//              arrive (a ticket) writes done = k
// against the same number of back-to-back launches of an empty kernel of the same shape on S.
// hipcc --offload-arch=gfx950 -O2 -o stream_value_probe stream_value_probe.hip && ./stream_value_probe [workgroups] [passes]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void resident(unsigned long long* go, unsigned long long* done, unsigned int* ticket, int n, int* gave_up) {
    for (int k = 1; k <= n; ++k) {
        if (threadIdx.x == 0) {
            long spins = 0;
            while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)k) {
                if (++spins > (1l << 24)) { *gave_up = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (*gave_up) return;
        if (threadIdx.x == 0) {
            const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (unsigned int)k * gridDim.x - 1u)
                __hip_atomic_store(done, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 512, N = argc > 2 ? atoi(argv[2]) : 2000;
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    unsigned long long *go = nullptr, *done = nullptr;
    unsigned int* ticket = nullptr;
    int* gave_up = nullptr;
    CK(hipExtMallocWithFlags((void**)&go, 8, hipMallocSignalMemory));
    CK(hipExtMallocWithFlags((void**)&done, 8, hipMallocSignalMemory));
    CK(hipMalloc((void**)&ticket, 4)); CK(hipMalloc((void**)&gave_up, 4));
    CK(hipMemset(ticket, 0, 4)); CK(hipMemset(gave_up, 0, 4));
    CK(hipMemset(go, 0, 8)); CK(hipMemset(done, 0, 8));
    hipStream_t S, I;
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&I, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    // baseline: back-to-back launches
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(empty, dim3(G), dim3(768), 0, S, (int*)nullptr);
        CK(hipStreamSynchronize(S));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("empty kernel, %d x 768 threads, back to back: %.2f us per launch\n", G, us);
    }
    hipLaunchKernelGGL(resident, dim3(G), dim3(768), 0, I, go, done, ticket, N, gave_up);
    CK(hipGetLastError());
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 1; k <= N; ++k) {
        CK(hipStreamWriteValue64(S, go, (unsigned long long)k, 0));
        CK(hipStreamWaitValue64(S, done, (unsigned long long)k, hipStreamWaitValueGte, ~0ull));
    }
    CK(hipStreamSynchronize(S));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    CK(hipStreamSynchronize(I));
    int gu = 0;
    CK(hipMemcpy(&gu, gave_up, 4, hipMemcpyDeviceToHost));
    printf("resident kernel, %d workgroups, go by hipStreamWriteValue64 / done by hipStreamWaitValue64: %.2f us per pass (gave up: %d)\n", G, us, gu);
    return 0;
}
