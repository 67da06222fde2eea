// solve.hip — K9: dense SPD camera system  S * da = E  (src/Bundle.cc:457-458,
// `Cholesky<>(mS).backsub(vE)`: TooN's unpivoted LDL^T reading the lower triangle).
//
// Blocked right-looking LDL^T, block size 32, ONE launch per block column k.  Every workgroup of
// step k re-factors the 32x32 diagonal block in LDS (cheap, avoids an extra launch + hand-off) and,
// in the SAME 32-step loop, carries along
//   - its panel blocks  X = A_ik * Lkk^-T  (so L_ik = X * D^-1),
//   - the forward substitution of the right-hand side block  z_k = Lkk^-1 * b_k.
// Workgroup roles in step k (rem = NB-k-1): [0] finaliser: writes Lkk, D_k, z_k;
// [1..rem] panel row i: writes L_ik and b_i -= L_ik z_k; [rest] trailing tile (i,j): A_ij -= X_i D^-1 X_j^T.
// The factor goes to a separate buffer L, so no workgroup reads a block another one writes in the
// same launch.  A final single-workgroup kernel does D^-1 and the backward substitution with L^T.
#include "bundle.h"

#define NB SOLVE_NB
#define LDP (NB + 1)   // LDS pitch in doubles (odd -> conflict-free column access)

__global__ void __launch_bounds__(256) ldlt_step_kernel(BaDev d, int k) {
    __shared__ double Akk[NB * LDP];
    __shared__ double Ai[NB * LDP];
    __shared__ double Aj[NB * LDP];
    __shared__ double Dk[NB];
    __shared__ double bz[NB];
    const int npad = d.npad, nblk = npad / NB, rem = nblk - k - 1;
    double* __restrict__ S = d.SE;
    double* __restrict__ E = d.SE + (size_t)npad * npad;
    const int tid = threadIdx.x;
    const int r = tid >> 3, g = tid & 7;   // row, column group (columns g, g+8, g+16, g+24)
    // role
    int role, bi = 0, bj = 0;
    const int wg = blockIdx.x;
    if (wg == 0)
        role = 0;
    else if (wg <= rem) {
        role = 1;
        bi = k + wg;
    } else {
        role = 2;
        int t = wg - rem - 1;   // tile index over (i,j), k < j <= i
        int ii = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
        while (ii * (ii + 1) / 2 > t) ii--;
        bi = k + 1 + ii;
        bj = k + 1 + (t - ii * (ii + 1) / 2);
    }
    const bool two = role == 2 && bi != bj;
    // load: lower triangle of A_kk (mirrored so the loop can read A[q][c] for q > c only), panels, rhs
    for (int q = g; q < NB; q += 8) {
        const int rr = r >= q ? r : q, cc = r >= q ? q : r;
        Akk[r * LDP + q] = S[(size_t)(k * NB + rr) * npad + k * NB + cc];
        if (role != 0) Ai[r * LDP + q] = S[(size_t)(bi * NB + r) * npad + k * NB + q];
        if (two) Aj[r * LDP + q] = S[(size_t)(bj * NB + r) * npad + k * NB + q];
    }
    if (tid < NB) bz[tid] = E[k * NB + tid];
    // the 32-step loop
    for (int c = 0; c < NB; c++) {
        __syncthreads();
        const double dc = Akk[c * LDP + c];
        const double inv = 1.0 / dc;
        // factor: trailing update of A_kk (lower part only)
        if (r > c) {
            const double lr = Akk[r * LDP + c] * inv;
            for (int q = g; q <= r; q += 8)
                if (q > c) Akk[r * LDP + q] -= lr * Akk[q * LDP + c];
            if (g == 0) bz[r] -= lr * bz[c];   // forward substitution of the rhs block
        }
        // panels: column c is final; eliminate it from the columns to its right
        if (role != 0) {
            const double xi = Ai[r * LDP + c];
            const double xj = two ? Aj[r * LDP + c] : 0.0;
            for (int q = g; q < NB; q += 8)
                if (q > c) {
                    const double l = Akk[q * LDP + c] * inv;
                    Ai[r * LDP + q] -= xi * l;
                    if (two) Aj[r * LDP + q] -= xj * l;
                }
        }
    }
    __syncthreads();
    if (tid < NB) Dk[tid] = Akk[tid * LDP + tid];
    __syncthreads();
    if (role == 0) {
        for (int q = g; q < NB; q += 8) {
            double v = 0.0;
            if (q < r)
                v = Akk[r * LDP + q] / Dk[q];
            else if (q == r)
                v = 1.0;
            d.L[(size_t)(k * NB + r) * npad + k * NB + q] = v;
        }
        if (tid < NB) {
            d.Dg[k * NB + tid] = Dk[tid];
            d.y[k * NB + tid] = bz[tid];
        }
    } else if (role == 1) {
        for (int q = g; q < NB; q += 8) d.L[(size_t)(bi * NB + r) * npad + k * NB + q] = Ai[r * LDP + q] / Dk[q];
        if (tid < NB) {
            double s = 0;
            for (int c = 0; c < NB; c++) s += (Ai[tid * LDP + c] / Dk[c]) * bz[c];
            E[bi * NB + tid] -= s;
        }
    } else {
        const double* __restrict__ Xj = two ? Aj : Ai;
        for (int q = g; q < NB; q += 8) {
            double s = 0;
#pragma unroll 8
            for (int c = 0; c < NB; c++) s += Ai[r * LDP + c] * (Xj[q * LDP + c] / Dk[c]);
            S[(size_t)(bi * NB + r) * npad + bj * NB + q] -= s;
        }
    }
}

// w = D^-1 z ; L^T x = w, blocked backwards.  One workgroup.
__global__ void __launch_bounds__(1024) ldlt_backward_kernel(BaDev d) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // npad doubles
    __shared__ double part[32][NB + 1];
    __shared__ double v[NB];
    const int npad = d.npad, nblk = npad / NB;
    const int tid = threadIdx.x;
    const int c = tid & 31, pr = tid >> 5;   // column within block, row partition (32 partitions)
    for (int k = nblk - 1; k >= 0; k--) {
        // s[c] = sum_{r > block k} L[r][k*NB + c] * x[r]
        double s = 0;
        for (int rr = (k + 1) * NB + pr; rr < npad; rr += 32) s += d.L[(size_t)rr * npad + k * NB + c] * xs[rr];
        part[pr][c] = s;
        __syncthreads();
        if (tid < NB) {
            double t = 0;
            for (int p = 0; p < 32; p++) t += part[p][tid];
            v[tid] = d.y[k * NB + tid] / d.Dg[k * NB + tid] - t;
        }
        __syncthreads();
        // unit upper-triangular solve Lkk^T x = v, right-looking from the last row
        for (int cc = NB - 1; cc >= 0; cc--) {
            if (tid < cc) v[tid] -= d.L[(size_t)(k * NB + cc) * npad + k * NB + tid] * v[cc];
            __syncthreads();
        }
        if (tid < NB) xs[k * NB + tid] = v[tid];
        __syncthreads();
    }
    for (int i = tid; i < npad; i += 1024) d.da[i] = xs[i];
}

int ba_solve(ptam_ctx* ctx, BaDev& d) {
    const int nblk = d.npad / NB;
    for (int k = 0; k < nblk; k++) {
        const int rem = nblk - k - 1;
        const int nwg = 1 + rem + rem * (rem + 1) / 2;
        hipLaunchKernelGGL(ldlt_step_kernel, dim3(nwg), dim3(256), 0, ctx->stream, d, k);
    }
    hipLaunchKernelGGL(ldlt_backward_kernel, dim3(1), dim3(1024), (size_t)d.npad * sizeof(double), ctx->stream, d);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}
