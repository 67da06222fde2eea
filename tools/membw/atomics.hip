// fp64 add-atomics from the whole chip onto a small set of addresses: what would summing the Schur kernel's ~600 partial tiles
// (48 x 48 doubles each, 28 tile pairs at the headline) by `global_atomic_add_f64` cost, instead of storing them and running
// schur_reduce_kernel?   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics atomics.hip -o atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) add_kernel(double* dst, int n_pairs, int elems) {
    // workgroup b flushes one partial tile of pair b % n_pairs: 256 threads x elems / 256 atomics
    double* t = dst + (size_t)(blockIdx.x % n_pairs) * elems;
    for (int e = threadIdx.x; e < elems; e += 256) atomicAdd(&t[e], 1.0 + e);
}
__global__ void __launch_bounds__(256) store_kernel(double* dst, int elems) {
    double* t = dst + (size_t)blockIdx.x * elems;
    for (int e = threadIdx.x; e < elems; e += 256) t[e] = 1.0 + e;
}
int main(int argc, char** argv) {
    const int partials = argc > 1 ? atoi(argv[1]) : 600, pairs = argc > 2 ? atoi(argv[2]) : 28, elems = 48 * 48;
    double *acc, *parts;
    CK(hipMalloc(&acc, (size_t)pairs * elems * 8));
    CK(hipMalloc(&parts, (size_t)partials * elems * 8));
    CK(hipMemset(acc, 0, (size_t)pairs * elems * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 50; i++) {
                if (mode == 0)
                    hipLaunchKernelGGL(add_kernel, dim3(partials), dim3(256), 0, 0, acc, pairs, elems);
                else
                    hipLaunchKernelGGL(store_kernel, dim3(partials), dim3(256), 0, 0, parts, elems);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%s: %d partial tiles of %d doubles onto %d tiles: %.2f us per launch (launch floor included)\n", mode == 0 ? "fp64 atomics" : "plain stores",
               partials, elems, pairs, 1e3 * best / 50);
    }
    return 0;
}
