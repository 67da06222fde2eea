// Does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=N: the first N dwords of the kernel arguments arrive in SGPRs with
// the wave instead of being fetched by its first scalar load) shorten a GUARDED kernel that leaves at once — the five launches of
// the speculative step prologue behind a rejected trial?  Two kernels that read a flag through a pointer and return: the pointer
// inside a 600-byte by-value struct (as BaDev carries d.sc: never preloaded), and as a leading argument (preloaded).
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=4 preload.hip -o preload
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { int a[140]; int* sc; double* out; };
__global__ void k_plain(Big d) { if (d.sc[3] == 0) return; d.out[threadIdx.x] = 1.0; }
__global__ void k_pre(const int* sc, Big d) { if (sc[3] == 0) return; d.out[threadIdx.x] = 1.0; }
__global__ void k_empty() {}
int main() {
    int* sc; double* out;
    hipMalloc(&sc, 64); hipMemset(sc, 0, 64); hipMalloc(&out, 4096);
    Big d{}; d.sc = sc; d.out = out;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int variant = 0; variant < 3; variant++)
        for (int rep = 0; rep < 3; rep++) {
            const int N = 2000;
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(245), dim3(256), 0, s);
                if (variant == 1) hipLaunchKernelGGL(k_plain, dim3(245), dim3(256), 0, s, d);
                if (variant == 2) hipLaunchKernelGGL(k_pre, dim3(245), dim3(256), 0, s, (const int*)sc, d);
            }
            hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            printf("%s: %.2f us per dependent launch (245 x 256 threads)\n", variant == 0 ? "empty kernel      " : variant == 1 ? "flag via the struct" : "flag via preloaded ", us);
        }
    return 0;
}
