// fp64 add-atomics with a fan-in: W workgroups add a 48 x 48 tile each onto one of T target tiles.  What does the fan-in per
// address cost when the adders of a target sit on ALL XCDs (target = block % T) or on ONE XCD (target = (block & 7) + 8 * k:
// block b is placed on XCD b % 8), at agent scope (the line's ownership moves between the XCDs' L2s) and at system scope (sc1:
// performed at the memory side)?  And a returning ticket (one word, every workgroup) at both scopes.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics atomics2.hip -o atomics2 && ./atomics2 [workgroups] [targets]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <bool SYS, bool ONE_XCD>
__global__ void __launch_bounds__(256) add_kernel(double* dst, int targets, int elems) {
    const int b = blockIdx.x;
    const int t = ONE_XCD ? (b & 7) + 8 * ((b >> 3) % (targets / 8)) : b % targets;
    double* p = dst + (size_t)t * elems;
    for (int e = threadIdx.x; e < elems; e += 256) {
        if (SYS)
            __hip_atomic_fetch_add(p + e, 1.0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else
            __hip_atomic_fetch_add(p + e, 1.0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// MODE 0: one agent-scope word; 1: one system-scope word; 2: a word per XCD (block & 7), agent scope, the last of an XCD adds to a common word
template <int MODE>
__global__ void __launch_bounds__(256) ticket_kernel(unsigned* t, unsigned per_xcd, unsigned* sink) {
    if (threadIdx.x == 0) {
        unsigned got;
        if (MODE == 0)
            got = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1)
            got = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else {
            got = __hip_atomic_fetch_add(t + 32 * (1 + (blockIdx.x & 7)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((got + 1) % per_xcd == 0) got = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (got == 0xffffffffu) sink[0] = got;
    }
}
__global__ void __launch_bounds__(256) wb_kernel(double* dst, int elems, int fence) {
    double* p = dst + (size_t)blockIdx.x * elems;
    for (int e = threadIdx.x; e < elems; e += 256) p[e] = 1.0 + e;
    if (fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 512, T = argc > 2 ? atoi(argv[2]) : 8, elems = 48 * 48;
    double* acc;
    unsigned* tk;
    CK(hipMalloc(&acc, (size_t)(T > W ? T : W) * elems * 8));
    CK(hipMalloc(&tk, 4096));
    CK(hipMemset(acc, 0, (size_t)(T > W ? T : W) * elems * 8));
    CK(hipMemset(tk, 0, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char* what) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; i++) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-58s %8.2f us per launch\n", what, 1e3 * best / 20);
    };
    printf("%d workgroups, %d targets (fan-in %d per address)\n", W, T, W / T);
    time([&] { hipLaunchKernelGGL((add_kernel<false, false>), dim3(W), dim3(256), 0, 0, acc, T, elems); }, "tile adds, agent scope, adders of a target on all XCDs");
    time([&] { hipLaunchKernelGGL((add_kernel<false, true>), dim3(W), dim3(256), 0, 0, acc, T, elems); }, "tile adds, agent scope, adders of a target on ONE XCD");
    time([&] { hipLaunchKernelGGL((add_kernel<true, false>), dim3(W), dim3(256), 0, 0, acc, T, elems); }, "tile adds, system scope, adders on all XCDs");
    time([&] { hipLaunchKernelGGL((add_kernel<true, true>), dim3(W), dim3(256), 0, 0, acc, T, elems); }, "tile adds, system scope, adders on ONE XCD");
    time([&] { hipLaunchKernelGGL(ticket_kernel<0>, dim3(W), dim3(256), 0, 0, tk, (unsigned)(W / 8), tk + 512); }, "returning ticket, one word, agent scope");
    time([&] { hipLaunchKernelGGL(ticket_kernel<1>, dim3(W), dim3(256), 0, 0, tk, (unsigned)(W / 8), tk + 512); }, "returning ticket, one word, system scope");
    time([&] { hipLaunchKernelGGL(ticket_kernel<2>, dim3(W), dim3(256), 0, 0, tk, (unsigned)(W / 8), tk + 512); }, "returning ticket, a word per XCD + common word");
    time([&] { hipLaunchKernelGGL(wb_kernel, dim3(W), dim3(256), 0, 0, acc, elems, 0); }, "plain tile stores");
    time([&] { hipLaunchKernelGGL(wb_kernel, dim3(W), dim3(256), 0, 0, acc, elems, 1); }, "plain tile stores + release fence (agent) per thread");
    return 0;
}
