// solve.hip — K9: dense SPD camera system  S * da = E  (src/Bundle.cc:457-458,
// `Cholesky<>(mS).backsub(vE)`: TooN's unpivoted LDL^T reading the lower triangle).
//
// Blocked right-looking LDL^T, block size 32, ONE launch per block column k.  Every workgroup of
// step k re-factors the 32x32 diagonal block in LDS (cheap; avoids an extra launch + hand-off) and,
// in the SAME loop, carries along
//   - its panel blocks  X = A_ik * Lkk^-T  (so L_ik = X * D^-1),
//   - the forward substitution of the right-hand side block  z_k = Lkk^-1 * b_k.
// The loop advances FOUR columns per iteration: the 4x4 pivot block is factored redundantly in
// registers by every thread (reciprocals by v_rcp_f64 + 2 Newton steps), so the sequential chain is
// 8 LDS round trips per block instead of 32.
// Workgroup roles in step k (rem = NB-k-1): [0] finaliser: writes Lkk, D_k, z_k;
// [1..rem] panel row i: writes L_ik and b_i -= L_ik z_k; [rest] trailing tile (i,j): A_ij -= X_i D^-1 X_j^T.
// The factor goes to a separate buffer L, so no workgroup reads a block another one writes in the
// same launch.  A final single-workgroup kernel does D^-1 and the backward substitution with L^T
// (diagonal blocks staged in LDS, the 32-step triangular chain in registers of one wave).
#include "bundle.h"

#define NB SOLVE_NB
#define LDP (NB + 1)        // LDS pitch in doubles (odd -> conflict-free column access)
#define TPB (NB * NB / 4)   // threads per workgroup: 4 columns of one row per thread (256 @ NB=32, 1024 @ NB=64)
#define GROUPS (NB / 4)     // column groups: thread (r, g) owns columns g, g+GROUPS, g+2*GROUPS, g+3*GROUPS

__device__ __forceinline__ double fast_rcp(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
}

// unit-lower 4x4 micro factor (l), pivots' reciprocals (i) of a symmetric 4x4 given by its lower triangle
struct Micro {
    double l10, l20, l21, l30, l31, l32;
    double i0, i1, i2, i3;
    double d1, d2, d3, u21, u31, u32;
};
__device__ __forceinline__ Micro micro_factor(double m00, double m10, double m11, double m20, double m21, double m22,
                                              double m30, double m31, double m32, double m33) {
    Micro f;
    f.i0 = fast_rcp(m00);
    f.l10 = m10 * f.i0;
    f.l20 = m20 * f.i0;
    f.l30 = m30 * f.i0;
    f.d1 = m11 - f.l10 * m10;
    f.i1 = fast_rcp(f.d1);
    f.u21 = m21 - f.l20 * m10;
    f.u31 = m31 - f.l30 * m10;
    f.l21 = f.u21 * f.i1;
    f.l31 = f.u31 * f.i1;
    f.d2 = m22 - f.l20 * m20 - f.l21 * f.u21;
    f.i2 = fast_rcp(f.d2);
    f.u32 = m32 - f.l30 * m20 - f.l31 * f.u21;
    f.l32 = f.u32 * f.i2;
    f.d3 = m33 - f.l30 * m30 - f.l31 * f.u31 - f.l32 * f.u32;
    f.i3 = fast_rcp(f.d3);
    return f;
}
// x = a * Lmicro^-T  (row substitution through the unit-lower micro factor)
__device__ __forceinline__ void micro_subst(const Micro& f, const double a[4], double x[4]) {
    x[0] = a[0];
    x[1] = a[1] - x[0] * f.l10;
    x[2] = a[2] - x[0] * f.l20 - x[1] * f.l21;
    x[3] = a[3] - x[0] * f.l30 - x[1] * f.l31 - x[2] * f.l32;
}

__global__ void __launch_bounds__(TPB) ldlt_step_kernel(BaDev d, int k) {
    extern __shared__ __attribute__((aligned(16))) double step_lds[];
    double* Akk = step_lds;
    double* Ai = Akk + NB * LDP;
    double* Aj = Ai + NB * LDP;
    double* iD = Aj + NB * LDP;
    double* bz = iD + NB;
    const int npad = d.npad, nblk = npad / NB, rem = nblk - k - 1;
    double* __restrict__ S = d.SE;
    double* __restrict__ E = d.SE + (size_t)npad * npad;
    const int tid = threadIdx.x;
    const int r = tid / GROUPS, g = tid % GROUPS;   // row, column group
    int role, bi = 0, bj = 0;
    const int wg = blockIdx.x;
    if (wg == 0)
        role = 0;
    else if (wg <= rem) {
        role = 1;
        bi = k + wg;
    } else {
        role = 2;
        const int t = wg - rem - 1;   // tile index over (i,j), k < j <= i
        int ii = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
        while (ii * (ii + 1) / 2 > t) ii--;
        bi = k + 1 + ii;
        bj = k + 1 + (t - ii * (ii + 1) / 2);
    }
    const bool pan = role != 0;
    const bool two = role == 2 && bi != bj;
    // load: A_kk mirrored from its lower triangle, panels, rhs
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const int q = g + GROUPS * jj;
        const int rr = r >= q ? r : q, cc = r >= q ? q : r;
        Akk[r * LDP + q] = S[(size_t)(k * NB + rr) * npad + k * NB + cc];
        if (pan) Ai[r * LDP + q] = S[(size_t)(bi * NB + r) * npad + k * NB + q];
        if (two) Aj[r * LDP + q] = S[(size_t)(bj * NB + r) * npad + k * NB + q];
    }
    if (tid < NB) bz[tid] = E[k * NB + tid];
#ifdef K7_TIMING
    const long long ts0 = (long long)__builtin_readcyclecounter();
#endif

    for (int c0 = 0; c0 < NB; c0 += 4) {
        __syncthreads();   // (A) previous panel's writes visible
        // ---- reads (original values of this panel) ----
        const Micro f = micro_factor(Akk[c0 * LDP + c0], Akk[(c0 + 1) * LDP + c0], Akk[(c0 + 1) * LDP + c0 + 1],
                                     Akk[(c0 + 2) * LDP + c0], Akk[(c0 + 2) * LDP + c0 + 1], Akk[(c0 + 2) * LDP + c0 + 2],
                                     Akk[(c0 + 3) * LDP + c0], Akk[(c0 + 3) * LDP + c0 + 1], Akk[(c0 + 3) * LDP + c0 + 2],
                                     Akk[(c0 + 3) * LDP + c0 + 3]);
        // multipliers of my (up to 4) columns q: lq = (row q of Akk through the micro factor) * D^-1
        double lq[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int q = g + GROUPS * jj;
            if (q > c0 + 3) {
                const double a[4] = {Akk[q * LDP + c0], Akk[q * LDP + c0 + 1], Akk[q * LDP + c0 + 2], Akk[q * LDP + c0 + 3]};
                double x[4];
                micro_subst(f, a, x);
                lq[jj][0] = x[0] * f.i0;
                lq[jj][1] = x[1] * f.i1;
                lq[jj][2] = x[2] * f.i2;
                lq[jj][3] = x[3] * f.i3;
            } else {
                lq[jj][0] = lq[jj][1] = lq[jj][2] = lq[jj][3] = 0;
            }
        }
        // my row through the micro factor, in each matrix
        double xk[4] = {0, 0, 0, 0}, xi[4] = {0, 0, 0, 0}, xj[4] = {0, 0, 0, 0};
        if (r > c0 + 3) {
            const double a[4] = {Akk[r * LDP + c0], Akk[r * LDP + c0 + 1], Akk[r * LDP + c0 + 2], Akk[r * LDP + c0 + 3]};
            micro_subst(f, a, xk);
        }
        if (pan) {
            const double a[4] = {Ai[r * LDP + c0], Ai[r * LDP + c0 + 1], Ai[r * LDP + c0 + 2], Ai[r * LDP + c0 + 3]};
            micro_subst(f, a, xi);
        }
        if (two) {
            const double a[4] = {Aj[r * LDP + c0], Aj[r * LDP + c0 + 1], Aj[r * LDP + c0 + 2], Aj[r * LDP + c0 + 3]};
            micro_subst(f, a, xj);
        }
        double z[4];
        {
            const double b4[4] = {bz[c0], bz[c0 + 1], bz[c0 + 2], bz[c0 + 3]};
            micro_subst(f, b4, z);   // same recurrence: z = Lmicro^-1 b
        }
        const double bzr = (g == 1 && r > c0 + 3) ? bz[r] : 0.0;
        __syncthreads();   // (B) every read of the old panel is done
        // ---- writes ----
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int q = g + GROUPS * jj;
            if (q > c0 + 3) {
                const double* l = lq[jj];
                if (q <= r) Akk[r * LDP + q] -= xk[0] * l[0] + xk[1] * l[1] + xk[2] * l[2] + xk[3] * l[3];
                if (pan) Ai[r * LDP + q] -= xi[0] * l[0] + xi[1] * l[1] + xi[2] * l[2] + xi[3] * l[3];
                if (two) Aj[r * LDP + q] -= xj[0] * l[0] + xj[1] * l[1] + xj[2] * l[2] + xj[3] * l[3];
            }
        }
        if (g == 0) {
            if (r > c0 + 3) {
#pragma unroll
                for (int m = 0; m < 4; m++) Akk[r * LDP + c0 + m] = xk[m];
            }
            if (pan) {
#pragma unroll
                for (int m = 0; m < 4; m++) Ai[r * LDP + c0 + m] = xi[m];
            }
            if (two) {
#pragma unroll
                for (int m = 0; m < 4; m++) Aj[r * LDP + c0 + m] = xj[m];
            }
        }
        if (g == 1 && r > c0 + 3)   // rhs rows below the panel: b_r -= L[r][c0..c0+3] . z
            bz[r] = bzr - (xk[0] * f.i0 * z[0] + xk[1] * f.i1 * z[1] + xk[2] * f.i2 * z[2] + xk[3] * f.i3 * z[3]);
        if (tid == 2) {
            // rows of the pivot block itself: D on the diagonal, undivided L*D below it
            Akk[(c0 + 1) * LDP + c0 + 1] = f.d1;
            Akk[(c0 + 2) * LDP + c0 + 1] = f.u21;
            Akk[(c0 + 2) * LDP + c0 + 2] = f.d2;
            Akk[(c0 + 3) * LDP + c0 + 1] = f.u31;
            Akk[(c0 + 3) * LDP + c0 + 2] = f.u32;
            Akk[(c0 + 3) * LDP + c0 + 3] = f.d3;
            bz[c0 + 1] = z[1];
            bz[c0 + 2] = z[2];
            bz[c0 + 3] = z[3];
        }
    }
    __syncthreads();
#ifdef K7_TIMING
    const long long ts1 = (long long)__builtin_readcyclecounter();
#endif
    if (tid < NB) iD[tid] = fast_rcp(Akk[tid * LDP + tid]);
    __syncthreads();
    if (role == 0) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int q = g + GROUPS * jj;
            double v = 0.0;
            if (q < r)
                v = Akk[r * LDP + q] * iD[q];
            else if (q == r)
                v = 1.0;
            d.L[(size_t)(k * NB + r) * npad + k * NB + q] = v;
        }
        if (tid < NB) {
            d.Dg[k * NB + tid] = Akk[tid * LDP + tid];
            d.y[k * NB + tid] = bz[tid];
        }
    } else if (role == 1) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int q = g + GROUPS * jj;
            d.L[(size_t)(bi * NB + r) * npad + k * NB + q] = Ai[r * LDP + q] * iD[q];
        }
        if (tid < NB) {
            double s = 0;
            for (int c = 0; c < NB; c++) s += (Ai[tid * LDP + c] * iD[c]) * bz[c];
            E[bi * NB + tid] -= s;
        }
    } else {
        // scale X_j by D^-1 once, in place (X_i when the tile is diagonal needs an unscaled copy: use Aj)
        double* XjD = Aj;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int q = g + GROUPS * jj;
            XjD[r * LDP + q] = (two ? Aj[r * LDP + q] : Ai[r * LDP + q]) * iD[q];
        }
        __syncthreads();
        double acc[4] = {0, 0, 0, 0};
#pragma unroll 8
        for (int c = 0; c < NB; c++) {
            const double a = Ai[r * LDP + c];
#pragma unroll
            for (int jj = 0; jj < 4; jj++) acc[jj] += a * XjD[(g + GROUPS * jj) * LDP + c];
        }
#pragma unroll
        for (int jj = 0; jj < 4; jj++) S[(size_t)(bi * NB + r) * npad + bj * NB + g + GROUPS * jj] -= acc[jj];
    }
#ifdef K7_TIMING
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && k == 2) {
        long long* o = d.dbg + (blockIdx.x == 0 ? 0 : 4);
        o[0] = ts0;
        o[1] = ts1;
        o[2] = (long long)__builtin_readcyclecounter();
        o[3] = role;
    }
#endif
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// w = D^-1 z ; L^T x = w, blocked backwards.  One workgroup of 1024 threads.
#define BW_PART (1024 / NB)   // row partitions of the GEMV part
__global__ void __launch_bounds__(1024) ldlt_backward_kernel(BaDev d, int cur) {
    extern __shared__ __attribute__((aligned(16))) double bw_lds[];
    const int npad = d.npad, nblk = npad / NB;
    double* xs = bw_lds;                       // npad
    double* part = xs + npad;                  // BW_PART x (NB + 1)
    double* Lk = part + BW_PART * (NB + 1);    // NB x LDP
    const int tid = threadIdx.x;
    const int c = tid % NB, pr = tid / NB;     // column within block, row partition
    for (int k = nblk - 1; k >= 0; k--) {
        // stage the diagonal block of L (independent of the running solution)
        for (int rr = pr; rr < NB; rr += BW_PART) Lk[rr * LDP + c] = d.L[(size_t)(k * NB + rr) * npad + k * NB + c];
        // s[c] = sum_{r > block k} L[r][k*NB + c] * x[r]
        double s = 0;
        for (int rr = (k + 1) * NB + pr; rr < npad; rr += BW_PART) s += d.L[(size_t)rr * npad + k * NB + c] * xs[rr];
        part[pr * (NB + 1) + c] = s;
        __syncthreads();
        if (tid < 64) {
            // unit upper-triangular solve Lkk^T x = v in registers of one wave (lanes 0..NB-1)
            double vt = 0;
            if (tid < NB) {
                double t = 0;
                for (int p = 0; p < BW_PART; p++) t += part[p * (NB + 1) + tid];
                vt = d.y[k * NB + tid] / d.Dg[k * NB + tid] - t;
            }
            for (int cc = NB - 1; cc > 0; cc--) {
                const double xv = readlane_f64(vt, cc);
                if (tid < cc) vt -= Lk[cc * LDP + tid] * xv;
            }
            if (tid < NB) xs[k * NB + tid] = vt;
        }
        __syncthreads();
    }
    for (int i = tid; i < npad; i += 1024) d.da[i] = xs[i];
    // trial poses  exp(da_j) * se3CfW  (src/Bundle.cc:496-501) and |da|^2, straight from LDS: saves the
    // separate pose-update launch
    for (int c2 = tid; c2 < d.C; c2 += 1024) {
        const double* T = d.pose[cur] + 12 * c2;
        double* Tn = d.pose[cur ^ 1] + 12 * c2;
        const int f = d.cam_free[c2];
        double Tl[12], o[12];
#pragma unroll
        for (int i = 0; i < 12; i++) Tl[i] = T[i];
        if (f < 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) Tn[i] = Tl[i];
        } else {
            double mu[6];
#pragma unroll
            for (int i = 0; i < 6; i++) mu[i] = xs[6 * f + i];
            se3_exp_mul(mu, Tl, o);
#pragma unroll
            for (int i = 0; i < 12; i++) Tn[i] = o[i];
        }
    }
    if (tid < 64) {
        double sq = 0;
        for (int i = tid; i < d.n; i += 64) sq += xs[i] * xs[i];
        sq = wave_sum_f64(sq);
        if (tid == 0) d.sc->sumsq_cam = sq;
    }
}

#define STEP_LDS_BYTES ((3 * NB * LDP + 2 * NB) * sizeof(double))
int ba_solve_init() {
    HIP_TRY(hipFuncSetAttribute((const void*)ldlt_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)STEP_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void*)ldlt_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return PTAM_OK;
}

int ba_solve(ptam_ctx* ctx, BaDev& d, int cur) {
    const int nblk = d.npad / NB;
    for (int k = 0; k < nblk; k++) {
        const int rem = nblk - k - 1;
        const int nwg = 1 + rem + rem * (rem + 1) / 2;
        hipLaunchKernelGGL(ldlt_step_kernel, dim3(nwg), dim3(TPB), STEP_LDS_BYTES, ctx->stream, d, k);
    }
    const size_t bw_bytes = ((size_t)d.npad + BW_PART * (NB + 1) + NB * LDP) * sizeof(double);
    hipLaunchKernelGGL(ldlt_backward_kernel, dim3(1), dim3(1024), bw_bytes, ctx->stream, d, cur);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}
