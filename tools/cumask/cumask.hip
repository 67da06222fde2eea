// cumask.hip — which physical CUs does a hipExtStreamCreateWithCUMask stream get on MI355X (8 XCDs x 32 CUs)?
// Each workgroup records (XCC_ID, SE_ID, CU_ID) of the CU it ran on; per mask the histogram over XCDs and the number of
// distinct CUs is printed.  hipcc -O3 --offload-arch=gfx950 cumask.hip -o cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) where_kernel(uint32_t* out) {
    // keep the CU busy for a while so that the dispatcher has to spread the grid
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000) {}
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
    }
}
int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int N = 4096;
    uint32_t* d;
    CK(hipMalloc(&d, N * 4));
    std::vector<uint32_t> h(N);
    struct M { const char* name; uint32_t w[8]; };
    std::vector<M> masks;
    masks.push_back({"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}});
    masks.push_back({"bits 0..31", {~0u, 0, 0, 0, 0, 0, 0, 0}});
    masks.push_back({"bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0}});
    masks.push_back({"bits = 0 mod 8", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}});
    masks.push_back({"bits 0..223", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0}});
    masks.push_back({"bits 32..255", {0, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}});
    for (auto& m : masks) {
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, 8, m.w));
        CK(hipMemsetAsync(d, 0xff, N * 4, s));
        hipLaunchKernelGGL(where_kernel, dim3(N), dim3(256), 0, s, d);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, N * 4, hipMemcpyDeviceToHost));
        int per_xcc[16] = {0};
        std::set<uint32_t> cus, percu[16];
        for (int i = 0; i < N; i++) {
            const uint32_t xcc = h[i] >> 16, hw = h[i] & 0xffff;
            const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[xcc & 15]++;
            cus.insert(xcc << 16 | se << 8 | sh << 4 | cu);
            percu[xcc & 15].insert(se << 8 | sh << 4 | cu);
        }
        printf("%-16s distinct CUs %3zu | workgroups per XCC:", m.name, cus.size());
        for (int x = 0; x < 8; x++) printf(" %4d(%2zu)", per_xcc[x], percu[x].size());
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    return 0;
}
