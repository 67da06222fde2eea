// tools/pipes/lat.hip — instruction latencies seen by ONE wave per SIMD on gfx950 (tools only): what a dependent chain of
// fp64 vector operations, of v_rcp_f64, of v_mfma_f64_16x16x4_f64, an LDS write -> read round trip and a 4-wave
// s_barrier cost in shader-clock cycles (s_memtime around N repetitions, one 256-thread workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define N 256
#define PIN() asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(u), "+v"(acc), "+v"(acc2))
#define now() ({ PIN(); long long t_ = (long long)__builtin_readcyclecounter(); PIN(); t_; })
__global__ void __launch_bounds__(256, 2) k(double* out, long long* cyc, double a, double b) {
    __shared__ double lds[2048];
    const int tid = threadIdx.x;
    double x = tid * 1e-3 + 1.0, y = x + 1.0, z = x + 2.0, u = x + 3.0;
    v4f64 acc = {x, y, z, u}, acc2 = {y, z, u, x};
    long long t0, t1;
    int q = 0;
#define REPORT() if (tid == 0) cyc[q] = t1 - t0; q++;
    // 1. dependent v_fma_f64 chain
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) x = fma(x, b, a);
    t1 = now();
    REPORT()
    // 2. two independent chains interleaved
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) { x = fma(x, b, a); y = fma(y, b, a); }
    t1 = now();
    REPORT()
    // 3. four independent chains
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) { x = fma(x, b, a); y = fma(y, b, a); z = fma(z, b, a); u = fma(u, b, a); }
    t1 = now();
    REPORT()
    // 4. dependent v_rcp_f64 chain
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) x = __builtin_amdgcn_rcp(x);
    t1 = now();
    REPORT()
    // 5. dependent MFMA chain (same accumulator)
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    t1 = now();
    REPORT()
    // 6. two independent MFMA chains
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    }
    t1 = now();
    REPORT()
    // 7. MFMA whose A operand depends on the previous result through one VALU op (MFMA -> VALU -> MFMA)
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, b, acc, 0, 0, 0);
        x = acc[0] * b;
    }
    t1 = now();
    REPORT()
    // 8. LDS write -> read of another lane's value (same wave, no barrier)
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) {
        lds[tid] = x;
        x = lds[tid ^ 1] + a;
    }
    t1 = now();
    REPORT()
    // 9. LDS write -> s_barrier -> read (4 waves)
    t0 = now();
#pragma unroll 8
    for (int i = 0; i < N; i++) {
        lds[tid + 256 * (i & 1)] = x;
        __syncthreads();
        x = lds[((tid + 64) & 255) + 256 * (i & 1)] + a;
    }
    t1 = now();
    REPORT()
    // 10. s_barrier alone
    t0 = now();
#pragma unroll 8
    for (int i = 0; i < N; i++) __syncthreads();
    t1 = now();
    REPORT()
    // 11. dependent v_mul_f64 / v_add_f64 alternating
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) { x = x * b; x = x + a; }
    t1 = now();
    REPORT()
    // 12. v_readlane of a fresh VALU result + use as SGPR operand
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int lo = __builtin_amdgcn_readlane((int)__double_as_longlong(x), 5);
        x = fma(x, b, (double)lo * 1e-300);
    }
    t1 = now();
    REPORT()
    // 13. dependent fma, all three operands in VGPRs
    double vb = b + tid * 1e-18, va = a + tid * 1e-18;
    asm volatile("" : "+v"(vb), "+v"(va));
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) x = fma(x, vb, va);
    t1 = now();
    REPORT()
    // 14. four interleaved chains, all-VGPR operands
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) { x = fma(x, vb, va); y = fma(y, vb, va); z = fma(z, vb, va); u = fma(u, vb, va); }
    t1 = now();
    REPORT()
    // 15. 12 LDS reads (b64, lane-addressed) issued back to back, then waited for (per group)
    {
        double r[12];
        t0 = now();
#pragma unroll 4
        for (int i = 0; i < N / 4; i++) {
#pragma unroll
            for (int j = 0; j < 12; j++) r[j] = lds[(tid * 4 + j * 64 + i) & 2047];
#pragma unroll
            for (int j = 0; j < 12; j++) x += r[j];
        }
        t1 = now();
        if (tid == 0) cyc[q] = (t1 - t0) * 4; q++;
    }
    // 16. MFMA -> ds_write of its result -> barrier -> ds_read -> MFMA operand (the loop's skeleton)
    t0 = now();
#pragma unroll 4
    for (int i = 0; i < N; i++) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, vb, acc, 0, 0, 0);
        lds[tid + 256 * (i & 1)] = acc[0];
        __syncthreads();
        x = lds[((tid + 64) & 255) + 256 * (i & 1)];
    }
    t1 = now();
    REPORT()
    // 17. the same with ~40 dependent fma between read and MFMA
    t0 = now();
#pragma unroll 2
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 40; j++) x = fma(x, vb, va);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, vb, acc, 0, 0, 0);
        lds[tid + 256 * (i & 1)] = acc[0];
        __syncthreads();
        x = lds[((tid + 64) & 255) + 256 * (i & 1)];
    }
    t1 = now();
    REPORT()
    out[blockIdx.x * 256 + tid] = x + y + z + u + acc[0] + acc[1] + acc[2] + acc[3] + acc2[0] + acc2[3];
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 8 * 4); hipMalloc(&cyc, 64 * 8);
    const char* names[] = {"dependent v_fma_f64", "2 interleaved fma chains (per pair)", "4 interleaved fma chains (per quad)", "dependent v_rcp_f64",
                           "dependent MFMA f64 16x16x4 (same acc)", "2 independent MFMA chains (per pair)", "MFMA -> v_mul -> MFMA (per pair)",
                           "LDS write -> read same wave", "LDS write -> barrier -> read (4 waves)", "s_barrier alone (4 waves)", "dependent mul+add (per pair)",
                           "fma -> readlane -> cvt -> fma", "dependent fma, 3 VGPR operands", "4 interleaved fma chains, VGPR operands (quad)", "12 LDS reads in flight + 12 adds (per group)", "MFMA -> ds_write -> barrier -> ds_read", "40 dep fma + MFMA -> ds_write -> barrier -> read"};
    for (int blocks : {1, 2}) {
        k<<<blocks, 256>>>(out, cyc, 1e-9, 1.0 + 1e-12);
        k<<<blocks, 256>>>(out, cyc, 1e-9, 1.0 + 1e-12);
        hipDeviceSynchronize();
        long long h[17];
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("%d workgroup(s) of 256 threads:\n", blocks);
        for (int i = 0; i < 17; i++) printf("  %-44s %7.1f cycles\n", names[i], (double)h[i] / N);
    }
    return 0;
}
