// tools/pipes/rcpacc.hip — accuracy of v_rcp_f64 and of its Newton refinements on gfx950 (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double y = __builtin_amdgcn_rcp(d);
    r0[i] = y;
    y = fma(fma(-d, y, 1.0), y, y);
    r1[i] = y;
    y = fma(fma(-d, y, 1.0), y, y);
    r2[i] = y;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> u(-300, 300);
    for (auto& v : x) v = ldexp(1.0 + (rng() >> 12) * 0x1p-52, (int)u(rng)) * ((rng() & 1) ? 1 : -1);
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        const long double t = 1.0L / (long double)x[i];
        e0 = fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
        e1 = fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
        e2 = fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
    }
    printf("max relative error: v_rcp_f64 %.3e (2^%.1f), + 1 Newton %.3e (%.2f ulp), + 2 Newton %.3e (%.2f ulp)\n", e0, log2(e0), e1, e1 / 0x1p-53,
           e2, e2 / 0x1p-53);
    return 0;
}
