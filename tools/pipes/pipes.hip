// Does v_mfma_f64_16x16x4_f64 overlap with vector fp64 FMAs on gfx950?  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NM, int NV>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
    v4f64 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3 + i;
    const double a = threadIdx.x * 1e-9, b = 1.0 + threadIdx.x * 1e-12;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (m < NM) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; v++) x[v % 8] = fma(x[v % 8], b, a);
        }
    }
    double s = 0;
    for (int m = 0; m < 4; m++) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM, int NV>
void run(const char* name, double* out, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<NM, NV><<<blocks, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<NM, NV><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s blocks/CU %d : %.1f ns per iteration (4 groups)\n", name, blocks / 256, ms * 1e6 / iters);
}
int main() {
    double* out; hipMalloc(&out, 8 * 256 * 1024);
    for (int blocks : {256, 512}) {
        run<4, 0>("4 MFMA", out, blocks);
        run<0, 16>("4x16 FMA", out, blocks);
        run<4, 16>("4 MFMA + 4x16 FMA interleaved", out, blocks);
        run<4, 8>("4 MFMA + 4x8 FMA interleaved", out, blocks);
        run<0, 8>("4x8 FMA", out, blocks);
    }
    return 0;
}
