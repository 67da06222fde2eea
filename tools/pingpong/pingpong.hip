// tools/pingpong/pingpong.hip — what a hand-off between two workgroups INSIDE one launch costs on gfx950 (tools only).
// Two workgroups of 256 threads (blocks 0 and P of a small grid; block b is observed to run on XCD b % 8) bounce a
// payload of B bytes back and forth N times:
//   sender:   payload by 16-byte write-through stores (global_store_dwordx4 ... sc1), s_waitcnt vmcnt(0), then the
//             sequence number into the flag word (relaxed agent-scope atomic store = sc1 store)
//   receiver: lane 0 polls the flag with relaxed agent-scope loads (sc1: served by L2, never by this CU's L1), a
//             workgroup barrier, then the payload by sc1 loads, every word checked against the sequence number
// — the placement-independent form of MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility".
// Prints the one-way cost (round trip / 2) for same-XCD and cross-XCD partners and payloads of 16 B, 1 KB and 8 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void st_plain(double* p, double a, double b) {
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"((__attribute__((ext_vector_type(2))) double){a, b}) : "memory");
}
__device__ __forceinline__ void st_wt(double* p, double a, double b) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"((__attribute__((ext_vector_type(2))) double){a, b}) : "memory");
}
__device__ __forceinline__ void ld_l2(const double* p, double& a, double& b) {
    __attribute__((ext_vector_type(2))) double v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    a = v.x, b = v.y;
}
template <bool PLAIN>
__global__ void __launch_bounds__(256) pp(double* buf, unsigned* flags, int partner, int bytes, int n, long long* out, int* bad) {
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner ? 1 : -1);
    if (me < 0) return;
    const int tid = threadIdx.x;
    double* mine = buf + me * 2048;          // what I send (8 KB at most)
    const double* theirs = buf + (me ^ 1) * 2048;
    unsigned* fl_out = flags + me * 64;
    unsigned* fl_in = flags + (me ^ 1) * 64;
    const int pairs = bytes / 16;            // 16-byte units
    int nbad = 0;
    long long t0 = 0;
    for (int it = 1; it <= n; it++) {
        if (it == 11 && tid == 0) t0 = (long long)__builtin_amdgcn_s_memrealtime();
        if (me == 0 || it > 1 || true) {
            if (me == 1) {   // wait for the ping first
                if (tid == 0)
                    while (__hip_atomic_load(fl_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
                __syncthreads();
                for (int u = tid; u < pairs; u += 256) {
                    double a, b;
                    ld_l2(theirs + 2 * u, a, b);
                    if (a != (double)it || b != (double)(it + u)) nbad++;
                }
            }
            for (int u = tid; u < pairs; u += 256) {
                if (PLAIN) st_plain(mine + 2 * u, (double)it, (double)(it + u)); else st_wt(mine + 2 * u, (double)it, (double)(it + u));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(fl_out, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (me == 0) {   // wait for the pong
                if (tid == 0)
                    while (__hip_atomic_load(fl_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
                __syncthreads();
                for (int u = tid; u < pairs; u += 256) {
                    double a, b;
                    ld_l2(theirs + 2 * u, a, b);
                    if (a != (double)it || b != (double)(it + u)) nbad++;
                }
            }
        }
    }
    if (tid == 0 && me == 0) out[0] = (long long)__builtin_amdgcn_s_memrealtime() - t0;
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    double* buf; unsigned* flags; long long* out; int* bad;
    hipMalloc(&buf, 2 * 2048 * 8); hipMalloc(&flags, 1024); hipMalloc(&out, 64); hipMalloc(&bad, 4);
    const int n = 2010;
    for (int partner : {8, 1, 4}) {
        for (int bytes : {16, 1024, 8192}) {
            hipMemset(flags, 0, 1024); hipMemset(bad, 0, 4); hipMemset(buf, 0, 2 * 2048 * 8);
            for (int plain = 0; plain < 2; plain++) {
                hipMemset(flags, 0, 1024); hipMemset(bad, 0, 4); hipMemset(buf, 0, 2 * 2048 * 8);
                if (plain) pp<true><<<16, 256>>>(buf, flags, partner, bytes, n, out, bad); else pp<false><<<16, 256>>>(buf, flags, partner, bytes, n, out, bad);
                hipDeviceSynchronize();
                long long t; int nb;
                hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
                printf("blocks 0 <-> %d (%s XCD), payload %5d B, %s stores: one way %.2f us  (%d stale words)\n", partner, partner % 8 == 0 ? "same" : "other", bytes,
                       plain ? "plain (L2 write-back)  " : "sc1 (write-through)", t / 100.0 / (n - 10) / 2, nb);
            }
        }
    }
    return 0;
}
