// Which XCD does block b of a dispatch land on?  Is it b % 8 whatever was dispatched before (other grid sizes, another queue)?
//   hipcc -O3 --offload-arch=gfx950 xccmap.hip -o xccmap && ./xccmap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void rec(unsigned* out, int n) {
    if (threadIdx.x == 0 && (int)blockIdx.x < n) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf;   // HW_REG_XCC_ID
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4096);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    unsigned h[64];
    const int grids[] = {16, 3, 16, 5, 16, 21, 16, 48, 16, 1001, 16};
    for (int q = 0; q < 2; q++) {
        hipStream_t s = q ? s2 : s1;
        for (int g : grids) {
            hipMemsetAsync(d, 0xff, 4096, s);
            hipLaunchKernelGGL(rec, dim3(g), dim3(64), 0, s, d, 16);
            hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            printf("queue %d grid %4d: XCC of blocks 0..%d:", q, g, (g < 16 ? g : 16) - 1);
            for (int i = 0; i < (g < 16 ? g : 16); i++) printf(" %u", h[i]);
            printf("\n");
        }
    }
    return 0;
}
