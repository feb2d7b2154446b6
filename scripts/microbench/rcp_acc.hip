// Accuracy of v_rcp_f64 / v_rsq_f64 / v_sqrt_f64 raw results on gfx950 (relative error vs host).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    r0[i] = r;
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    r1[i] = r;
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    r2[i] = r;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), a(n), b(n), c(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (int i = 0; i < n; ++i) h[i] = std::exp(u(g) * 60.0 - 20.0) * (1.0 + u(g));
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)h[i];
        m0 = std::max(m0, (double)fabsl(((long double)a[i] - t) / t));
        m1 = std::max(m1, (double)fabsl(((long double)b[i] - t) / t));
        m2 = std::max(m2, (double)fabsl(((long double)c[i] - t) / t));
    }
    printf("v_rcp_f64 max rel err: raw %.3e, 1 NR %.3e, 2 NR %.3e\n", m0, m1, m2);
    return 0;
}
