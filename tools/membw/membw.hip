// HBM bandwidth probe (tools only): write / read / copy of N bytes with dwordx4 accesses, timed with HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(512) wr_planes(double2* __restrict__ W, int M) {   // K7's W pattern: 9 planes
    const int m = blockIdx.x * 512 + threadIdx.x;
    if (m < M)
#pragma unroll
        for (int q = 0; q < 9; q++) W[(size_t)q * M + m] = make_double2(m, q);
}
__global__ void __launch_bounds__(512) wr_lin(double2* __restrict__ W, size_t n) {
    for (size_t i = blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) W[i] = make_double2(1.0, 2.0);
}
__global__ void __launch_bounds__(512) rd_lin(const double2* __restrict__ W, size_t n, double* out) {
    double a = 0;
    for (size_t i = blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) { const double2 t = W[i]; a += t.x + t.y; }
    if (a == 123.456) out[0] = a;
}
__global__ void __launch_bounds__(512) cp_lin(const double2* __restrict__ A, double2* __restrict__ B, size_t n) {
    for (size_t i = blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) B[i] = A[i];
}
__global__ void __launch_bounds__(512) wr_planes_sc1(double2* __restrict__ W, int M) {   // the same, write-through
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int m = blockIdx.x * 512 + threadIdx.x;
    if (m < M)
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const d2 v = {(double)m, (double)q};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(&W[(size_t)q * M + m]), "v"(v) : "memory");
        }
}
__global__ void __launch_bounds__(512) nothing(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 800000;
    const size_t n = (size_t)M * 9, bytes = n * 16;
    double2 *A, *B; double* o;
    hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&o, 8);
    hipMemset(A, 0, bytes); hipMemset(B, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto fn, double traffic) {
        for (int i = 0; i < 5; i++) fn();
        float tot = 0;
        for (int i = 0; i < 20; i++) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms; }
        printf("%-28s M=%d  %.2f us  %.2f TB/s\n", name, M, tot / 20 * 1e3, traffic / (tot / 20 * 1e-3) / 1e12);
    };
    timeit("write 9 planes (1 meas/thr)", [&] { wr_planes<<<(M + 511) / 512, 512>>>(A, M); }, (double)bytes);
    timeit("write linear grid=2048", [&] { wr_lin<<<2048, 512>>>(A, n); }, (double)bytes);
    timeit("write linear grid=512", [&] { wr_lin<<<512, 512>>>(A, n); }, (double)bytes);
    timeit("read linear grid=2048", [&] { rd_lin<<<2048, 512>>>(A, n, o); }, (double)bytes);
    timeit("copy linear grid=2048", [&] { cp_lin<<<2048, 512>>>(A, B, n); }, 2.0 * bytes);
    timeit("empty-ish (M=64)", [&] { wr_planes<<<1, 512>>>(A, 64); }, 0.0);
    // dependent back-to-back launches in one stream: the per-launch floor without event overhead
    auto bracket = [&](const char* name, auto fn, double traffic) {
        for (int i = 0; i < 5; i++) fn();
        const int N = 100;
        hipEventRecord(e0);
        for (int i = 0; i < N; i++) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-36s bracket of %d: %.2f us / launch  %.2f TB/s\n", name, N, ms / N * 1e3, traffic / (ms / N * 1e-3) / 1e12);
    };
    bracket("empty, 1 workgroup", [&] { nothing<<<1, 512>>>(nullptr); }, 0.0);
    bracket("empty, 489 workgroups", [&] { nothing<<<489, 512>>>(nullptr); }, 0.0);
    bracket("empty, 4096 workgroups", [&] { nothing<<<4096, 512>>>(nullptr); }, 0.0);
    const int Mh = 250000;
    bracket("write 9 planes M=250000 plain", [&] { wr_planes<<<(Mh + 511) / 512, 512>>>(A, Mh); }, 144.0 * Mh);
    bracket("write 9 planes M=250000 sc1", [&] { wr_planes_sc1<<<(Mh + 511) / 512, 512>>>(A, Mh); }, 144.0 * Mh);
    bracket("write 9 planes M=800000 plain", [&] { wr_planes<<<(800000 + 511) / 512, 512>>>(A, 800000); }, 144.0 * 800000);
    bracket("write 9 planes M=800000 sc1", [&] { wr_planes_sc1<<<(800000 + 511) / 512, 512>>>(A, 800000); }, 144.0 * 800000);
    return 0;
}
