// atan_cr (ptam_cg_amd/csrc/atan_cr.h) on the device over the doubles of a file: atan_cr_check in.bin out.bin
//   hipcc -O3 --offload-arch=gfx950 -I../../ptam_cg_amd/csrc atan_cr_check.hip -o atan_cr_check   (driver: check.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "atan_cr.h"
__global__ void k(const double* x, double* y, double* y2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        y[i] = atan_cr(x[i]);
        y2[i] = atan(x[i]);
    }
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END);
    const long n = ftell(f) / 8;
    fseek(f, 0, SEEK_SET);
    std::vector<double> h(n), o(2 * n);
    if (fread(h.data(), 8, n, f) != (size_t)n) return 1;
    fclose(f);
    double *dx, *dy, *dy2;
    hipMalloc(&dx, n * 8);
    hipMalloc(&dy, n * 8);
    hipMalloc(&dy2, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, dy2, (int)n);
    hipMemcpy(o.data(), dy, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o.data() + n, dy2, n * 8, hipMemcpyDeviceToHost);
    f = fopen(argv[2], "wb");
    fwrite(o.data(), 8, 2 * n, f);
    fclose(f);
    return 0;
}
