// Does a kernel launched with hipExtAnyOrderLaunch start before its predecessor in the SAME stream has finished on gfx950, and
// what does a dependent "kernel boundary" cost when the dependency is a completion counter the successor spins on instead of the
// queue's barrier bit?   hipcc -O3 --offload-arch=gfx950 anyorder.hip -o anyorder && ./anyorder [blocks] [threads] [work_ns]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now() { return wall_clock64(); }   // 100 MHz

// one link of a chain: wait for link k-1's counter (FLAGS), work for `ticks`, raise my counter
#ifndef SLOTS
#define SLOTS 16
#endif
template <bool FLAGS>
__global__ void __launch_bounds__(1024) link_kernel(unsigned* cnt, int k, unsigned expect, unsigned long long ticks, unsigned long long* stamps, double* sink) {
    // completion counters: SLOTS per link, 128 bytes apart (256 same-address device-scope atomics cost 5 us: ~20 ns each);
    // block b raises slot b % SLOTS, the waiter's lanes 0..SLOTS-1 poll one slot each
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamps) stamps[2 * k] = now();
    if (FLAGS && k > 0 && threadIdx.x < 64) {
        const int sl = threadIdx.x;
        const unsigned want = sl < SLOTS ? (expect + SLOTS - 1 - sl) / SLOTS : 0;
        unsigned long long t0 = now();
        for (;;) {
            const unsigned v = sl < SLOTS ? __hip_atomic_load(&cnt[((k - 1) * SLOTS + sl) * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            if (__ballot(v < want) == 0) break;
            if (now() - t0 > 200000000ull) break;   // 2 s
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    const unsigned long long t0 = now();
    double a = threadIdx.x;
    while (now() - t0 < ticks) a = a * 1.0000001 + 1e-9;
    if (a == 12345.678) sink[0] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (stamps) atomicMax(&stamps[2 * k + 1], now());   // (end of the link: the latest block)
        __hip_atomic_fetch_add(&cnt[(k * SLOTS + blockIdx.x % SLOTS) * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 256;
    const unsigned long long ticks = (argc > 3 ? atoi(argv[3]) : 1000) / 10;
    const int K = 200;
    unsigned* cnt;
    unsigned long long* stamps;
    double* sink;
    CK(hipMalloc(&cnt, K * SLOTS * 128));
    CK(hipMalloc(&stamps, K * 16));
    CK(hipMalloc(&sink, 8));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(2 * K);
    for (int mode = 0; mode < 3; mode++) {   // 0: ordered launches, no flags; 1: ordered launches + flags; 2: any-order launches + flags
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            unsigned long long* stp = rep == 5 ? stamps : nullptr;   // (the stamps' same-address atomicMax costs microseconds: last repetition only, not timed)
            CK(hipMemsetAsync(cnt, 0, K * SLOTS * 128, st));
            CK(hipMemsetAsync(stamps, 0, K * 16, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < K; k++) {
                if (mode == 0)
                    hipLaunchKernelGGL(link_kernel<false>, dim3(blocks), dim3(threads), 0, st, cnt, k, (unsigned)blocks, ticks, stp, sink);
                else
                    hipExtLaunchKernelGGL(link_kernel<true>, dim3(blocks), dim3(threads), 0, st, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0, cnt, k,
                                          (unsigned)blocks, ticks, stp, sink);
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best && rep < 5) best = ms;
        }
        CK(hipMemcpy(h.data(), stamps, K * 16, hipMemcpyDeviceToHost));
        // overlap: start of link k against the end of link k-1 (negative = the successor was on the chip before the predecessor finished)
        double sum = 0;
        int neg = 0;
        for (int k = 1; k < K; k++) {
            const double d = ((double)h[2 * k] - (double)h[2 * k - 1]) * 10.0;
            sum += d;
            neg += d < 0;
        }
        printf("mode %d (%s): %.2f us per link (work %.2f us), start(k) - end(k-1) mean %.0f ns, %d of %d links started early\n", mode,
               mode == 0 ? "ordered" : mode == 1 ? "ordered + flags" : "any-order + flags", best * 1000.0 / K, ticks * 0.01, sum / (K - 1), neg, K - 1);
    }
    return 0;
}
