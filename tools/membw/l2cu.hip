// tools/membw/l2cu.hip — how fast can ONE compute unit pull L2-resident data?  (round 5: the question behind the camera
// solve's backward substitution, which streams the 360 KB factor through the one CU of its workgroup.)
//   hipcc -O3 --offload-arch=gfx950 l2cu.hip -o l2cu ; ./l2cu
// One workgroup (1, 2, 4, 8, 16 waves) reads a 384 KB buffer that another kernel has just written (so it sits in the L2,
// as the factor does), 8 or 16 bytes per lane, agent-scope (sc1, never served by the L1) or plain loads; cycles by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill(double* p, size_t n) {
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0 + 1e-9 * i;
}
template <int W, bool SC1>
__global__ void rd(const double* p, size_t n, double* out, long long* cyc) {   // W = doubles per lane and load (1 or 2)
    const int tid = threadIdx.x, nt = blockDim.x;
    double acc = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    constexpr int U = 8;
    for (size_t base = 0; base + (size_t)U * nt * W <= n; base += (size_t)U * nt * W) {
        double v[U][W];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const double* q = p + base + ((size_t)u * nt + tid) * W;
            if (W == 1) {
                if (SC1) v[u][0] = __longlong_as_double(__hip_atomic_load((const long long*)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                else v[u][0] = *q;
            } else {
                double2 t;
                if (SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(q) : "memory");
                else t = *(const double2*)q;
                v[u][0] = t.x, v[u][1] = t.y;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            for (int i = 0; i < W; i++) acc += v[u][i];
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    out[tid] = acc;
    if (tid == 0) cyc[0] = t1 - t0;
}
int main() {
    const size_t n = 48 * 1024;   // doubles: 384 KB
    double *p, *out;
    long long* cyc;
    CK(hipMalloc(&p, n * 8));
    CK(hipMalloc(&out, 1024 * 8));
    CK(hipMalloc(&cyc, 8));
    printf("one workgroup reads %zu KB out of the L2: cycles, bytes per cycle\n", n * 8 / 1024);
    for (int waves : {1, 2, 4, 8, 16}) {
        for (int var = 0; var < 3; var++) {   // (x4 sc1 is serialised by its inline wait and left out)
            long long best = 1ll << 60;
            for (int rep = 0; rep < 5; rep++) {
                fill<<<64, 256>>>(p, n);
                if (var == 0) rd<1, true><<<1, 64 * waves>>>(p, n, out, cyc);
                if (var == 1) rd<1, false><<<1, 64 * waves>>>(p, n, out, cyc);
                if (var == 2) rd<2, false><<<1, 64 * waves>>>(p, n, out, cyc);
                CK(hipDeviceSynchronize());
                long long c;
                CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
                if (c < best) best = c;
            }
            const char* nm[] = {"8 B/lane sc1  ", "8 B/lane plain", "16 B/lane plain"};
            printf("  %2d waves, %s: %8lld cycles  %6.1f B/cycle\n", waves, nm[var], best, (double)(n * 8) / best);
        }
    }
    return 0;
}
