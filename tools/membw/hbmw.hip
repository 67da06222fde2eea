// hbmw.hip — what a streaming kernel of K7's shape can reach on MI355X: W bytes written (16-byte stores) + R bytes read per launch,
// 200 launches back to back (HIP events), for a few grid shapes.  hipcc -O3 --offload-arch=gfx950 hbmw.hip -o hbmw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(1024) stream_kernel(const double2* __restrict__ in, double2* __restrict__ out, size_t n_in, size_t n_out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, T = (size_t)gridDim.x * blockDim.x;
    double2 acc = make_double2(0, 0);
    for (size_t i = t; i < n_in; i += T) { const double2 v = in[i]; acc.x += v.x; acc.y += v.y; }
    for (size_t i = t; i < n_out; i += T) out[i] = make_double2(acc.x + (double)i, acc.y);
}
int main(int argc, char** argv) {
    const size_t wbytes = (argc > 1 ? atof(argv[1]) : 39.7) * 1e6, rbytes = (argc > 2 ? atof(argv[2]) : 10.7) * 1e6;
    double2 *in, *out;
    CK(hipMalloc(&in, rbytes + 64)); CK(hipMalloc(&out, wbytes + 64));
    CK(hipMemset(in, 0, rbytes)); CK(hipMemset(out, 0, wbytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int shapes[][2] = {{512, 1024}, {256, 1024}, {1024, 1024}, {2048, 256}, {8192, 256}, {512, 512}};
    for (auto& s : shapes) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 200; i++) hipLaunchKernelGGL(stream_kernel, dim3(s[0]), dim3(s[1]), 0, 0, in, out, rbytes / 16, wbytes / 16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("grid %5d x %4d: %.2f us per launch, %.2f TB/s (write %.1f MB + read %.1f MB)\n", s[0], s[1], ms * 5.0, (wbytes + rbytes) / (ms * 5e-6) / 1e12, wbytes / 1e6, rbytes / 1e6);
        }
    }
    return 0;
}
