// solve.hip — K9: dense SPD camera system  S * da = E  (src/Bundle.cc:457-458,
// `Cholesky<>(mS).backsub(vE)`: TooN's unpivoted LDL^T reading the lower triangle).
//
// Blocked right-looking LDL^T, block size 32, ONE launch per block column k.  Every workgroup of
// step k re-factors the 32x32 diagonal block (cheap; avoids an extra launch + hand-off) and, in the
// SAME loop, carries along
//   - its panel blocks  X = A_ik * Lkk^-T  (so L_ik = X * D^-1),
//   - the forward substitution of the right-hand side block  z_k = Lkk^-1 * b_k.
// The sequential chain is what bounds a step, so it is kept as short as the data flow allows:
//   - thread (r, g) of the 256 owns row r, columns g, g+8, g+16, g+24 (cyclic, so the remaining work
//     per thread shrinks evenly) of every block the workgroup carries, IN REGISTERS, for the whole step;
//   - the loop advances FOUR columns per iteration.  Only the panel of the iteration (32 rows x 4
//     columns per block) goes through LDS, double buffered, so an
//     iteration is ONE barrier and one LDS round trip: read panel -> factor the 4x4 pivot block
//     redundantly in registers (reciprocals by v_rcp_f64 + 2 Newton steps) -> substitute my rows ->
//     rank-4 update of my strips -> the owners of the next panel publish it.
// Workgroup roles in step k (rem = min(blocks below k, block bandwidth of S): outside the band S, and therefore L, is zero): [0] finaliser: writes D_k, z_k and — by carrying the
// identity as its panel block — Lkk^-T, which is all the backward substitution needs of the diagonal
// block; [1..rem] panel row i: writes L_ik and b_i -= L_ik z_k; [rest] trailing tile (i,j):
// A_ij -= X_i D^-1 X_j^T.  The factor goes to a separate buffer L, so no workgroup reads a block
// another one writes in the same launch.
// A final single-workgroup kernel does D^-1 and the backward substitution with L^T, right-looking:
// per block a 32x32 mat-vec with Lkk^-T, then every pending block below is updated in parallel.
#include "bundle.h"

#define NB SOLVE_NB
#define LDP (NB + 1)    // LDS pitch in doubles (odd -> conflict-free column access)
#define STRIPS 8                  // column residues: thread (r, g) owns columns g + STRIPS*jj
#define CPT (NB / STRIPS)         // columns per thread
#define TPB (NB * STRIPS)         // 256 threads (512 = two waves per SIMD measured slower: the barrier grows, the chain does not shrink)
#define NITER (NB / 4)
static_assert(NB == 32, "the strip decomposition below is written for 32x32 blocks");

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
    double d0, d1, d2, d3, u10, u20, u30, u21, u31, u32;
};
__device__ __forceinline__ Micro micro_factor(double m00, double m10, double m11, double m20, double m21, double m22,
                                              double m30, double m31, double m32, double m33) {
    Micro f;
    f.d0 = m00;
    f.u10 = m10;
    f.u20 = m20;
    f.u30 = m30;
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

struct StepLds {
    double P[2][3][NB][4];   // the iteration's panel (raw strips of A_kk, A_i, A_j), double buffered
    double Pb[2][4];         // rhs entries of the pivot rows
    double Xi[NB * LDP];     // closing phase: X_i (and X_j D^-1) tiles
    double Xj[NB * LDP];
    double iD[NB], Dv[NB], z[NB];
};

__global__ void __launch_bounds__(TPB) ldlt_step_kernel(BaDev d, int k) {
    if (k == 0) { TL_MARK(d, 9) }
    __shared__ StepLds s;
    const int npad = d.npad, nblk = npad / NB, band = se_band(d), rem = min(nblk - k - 1, band);   // (below the band column k holds zeros: not stored)
    double* __restrict__ S = d.SE;
    double* __restrict__ E = se_E(d);
    const int tid = threadIdx.x;
    const int r = tid / STRIPS, g = tid % STRIPS;   // row, column residue (columns g + 8*jj)
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
    const bool two = role == 2 && bi != bj;
    // ---- load my columns q = g + 8*jj (A_kk: only its lower triangle is ever read) ----
    double akk[CPT], ai[CPT], aj[CPT];
#pragma unroll
    for (int jj = 0; jj < CPT; jj++) {
        const int q = g + STRIPS * jj;
        aj[jj] = 0.0;
        akk[jj] = S[se_blk(k, k, band) + r * NB + q];
        if (role == 0)   // the identity rides along as the panel block: its X is Lkk^-T
            ai[jj] = (q == r) ? 1.0 : 0.0;
        else
            ai[jj] = S[se_blk(bi, k, band) + r * NB + q];
        if (two) aj[jj] = S[se_blk(bj, k, band) + r * NB + q];
    }
    double br = (g == STRIPS - 1) ? E[k * NB + r] : 0.0;   // thread (r, 7) carries b_r
    // operands of the closing read-modify-writes, fetched now so that the step does not end on a load
    const int mw = tid >> 6, mlane = tid & 63;
    const int qi = mw >> 1, qj = mw & 1;   // role 2: wave w owns the 16x16 output quadrant (w>>1, w&1)
    double sij[4] = {0, 0, 0, 0};
    if (role == 2) {
#pragma unroll
        for (int v = 0; v < 4; v++)
            sij[v] = S[se_blk(bi, bj, band) + (16 * qi + (mlane >> 4) + 4 * v) * NB + 16 * qj + (mlane & 15)];
    }
    const double e_old = (role == 1 && g == STRIPS - 1) ? E[bi * NB + r] : 0.0;
#ifdef K7_TIMING
    const long long ts0 = (long long)__builtin_readcyclecounter();
#endif
    // publish panel 0 (columns 0..3: threads g < 4, register 0)
    if (g < 4) {
        s.P[0][0][r][g] = akk[0];
        s.P[0][1][r][g] = ai[0];
        s.P[0][2][r][g] = aj[0];
    }
    if (g == STRIPS - 1 && r < 4) s.Pb[0][r] = br;
    // the last block column ends in identity padding (npad - n rows): its pivots are 1, its multipliers 0 and its
    // right-hand side 0, so the iterations that would only walk over padding are skipped
    const int ncols = min(NB, d.n - k * NB), niter = (ncols + 3) / 4;
    if (tid < NB) {
        s.iD[tid] = 1.0;
        s.Dv[tid] = 1.0;
        s.z[tid] = 0.0;
    }
    __syncthreads();

    // Fully unrolled: the panel of iteration t is register t/2 of the threads with g/4 == t%2, so every
    // register index below is a compile-time constant.
    // Software pipelined around the barrier.  The chain that bounds a step is
    //     panel in LDS -> micro factor -> my rows through it -> the NEXT panel's columns updated -> published -> barrier,
    // so only that much happens before the barrier; the rank-4 update of all my other columns, the final values of the
    // panel columns and the bookkeeping follow AFTER it, behind the LDS reads of the next iteration's operands — work
    // that used to sit on the chain now fills the read latency (one wave per SIMD: nothing else would).
    struct IterOps {
        double pv[10];                                 // pivot block, lower triangle 00 10 11 20 21 22 30 31 32 33
        double rk[4], ri[4], rj[4], b4[4], cq[CPT][4];   // my row of the three panels, rhs pivots, multiplier rows of my columns
    };
    auto load_ops = [&](int t, IterOps& o) {
        const int c0 = 4 * t, pb = t & 1, jp = c0 / STRIPS;
        const double(*Pk)[4] = s.P[pb][0];
        const double(*Pi)[4] = s.P[pb][1];
        const double(*Pj)[4] = s.P[pb][2];
        int q = 0;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int n = 0; n <= m; n++) o.pv[q++] = Pk[c0 + m][n];
#pragma unroll
        for (int n = 0; n < 4; n++) {
            o.rk[n] = Pk[r][n];
            o.ri[n] = Pi[r][n];
            o.rj[n] = Pj[r][n];
            o.b4[n] = s.Pb[pb][n];
        }
#pragma unroll
        for (int jj = 0; jj < CPT; jj++)
#pragma unroll
            for (int n = 0; n < 4; n++) o.cq[jj][n] = jj >= jp ? Pk[g + STRIPS * jj][n] : 0.0;
    };
    IterOps cur;
    load_ops(0, cur);
#pragma unroll
    for (int t = 0; t < NITER; t++) {
        if (t >= niter) break;   // (uniform; the register indices below stay compile-time constants)
        const int c0 = 4 * t, pb = t & 1, jp = c0 / STRIPS, gh = (c0 % STRIPS) / 4;
        const bool has_next = t + 1 < NITER;
        const int jn = (c0 + 4) / STRIPS, ghn = ((c0 + 4) % STRIPS) / 4;   // register / thread half of the next panel
        const bool below = r > c0 + 3;
        const Micro f = micro_factor(cur.pv[0], cur.pv[1], cur.pv[2], cur.pv[3], cur.pv[4], cur.pv[5], cur.pv[6], cur.pv[7], cur.pv[8],
                                     cur.pv[9]);
        // my rows of the panel through the micro factor
        // (selects, not branches, on the chain: with one wave per SIMD the scheduler's freedom to interleave the
        //  independent substitutions is the only latency hiding there is, and an exec-mask branch fences it)
        double xk[4], xi[4], xj[4];
        micro_subst(f, cur.rk, xk);
#pragma unroll
        for (int n = 0; n < 4; n++) xk[n] = below ? xk[n] : 0.0;
        micro_subst(f, cur.ri, xi);
        micro_subst(f, cur.rj, xj);   // (zeros unless the workgroup carries a second panel block)
        // rank-4 update of one of my columns right of the panel: the multipliers of column q are row q of the panel
        auto update_col = [&](int jj) {
            if (jj == jp && (c0 % STRIPS) != 0) return;   // (compile time: none of my columns in register jp lies right of the panel)
            const bool on = jj > jp || g > 3;             // register jp, first half of the block: columns g > 3 only
            double x[4];
            micro_subst(f, cur.cq[jj], x);
            const double l0 = x[0] * f.i0, l1 = x[1] * f.i1, l2 = x[2] * f.i2, l3 = x[3] * f.i3;
            const double nk = akk[jj] - (xk[0] * l0 + xk[1] * l1 + xk[2] * l2 + xk[3] * l3);
            const double ni = ai[jj] - (xi[0] * l0 + xi[1] * l1 + xi[2] * l2 + xi[3] * l3);
            const double nj = aj[jj] - (xj[0] * l0 + xj[1] * l1 + xj[2] * l2 + xj[3] * l3);
            akk[jj] = on ? nk : akk[jj];
            ai[jj] = on ? ni : ai[jj];
            aj[jj] = on ? nj : aj[jj];
        };
        // ---- on the chain: the next panel's columns and the right-hand side, then publish ----
        double z[4];
        micro_subst(f, cur.b4, z);   // z = Lmicro^-1 b (same recurrence), rows below take b_r -= L[r][c0..c0+3] . z
        {
            const double nb = br - (xk[0] * f.i0 * z[0] + xk[1] * f.i1 * z[1] + xk[2] * f.i2 * z[2] + xk[3] * f.i3 * z[3]);
            br = (g == STRIPS - 1 && below) ? nb : br;
        }
        if (has_next) {
            update_col(jn);
            if ((g >> 2) == ghn) {
                s.P[pb ^ 1][0][r][g & 3] = akk[jn];
                s.P[pb ^ 1][1][r][g & 3] = ai[jn];
                s.P[pb ^ 1][2][r][g & 3] = aj[jn];
            }
            if (g == STRIPS - 1 && r >= c0 + 4 && r < c0 + 8) s.Pb[pb ^ 1][r - c0 - 4] = br;
        }
        __syncthreads();
        // ---- off the chain: next operands requested, then everything else of this iteration ----
        IterOps nxt;
        if (has_next) load_ops(t + 1, nxt);
#pragma unroll
        for (int jj = jp; jj < CPT; jj++)
            if (!(has_next && jj == jn)) update_col(jj);
        if ((g >> 2) == gh) {
            // my column jp IS panel column n: it takes its final value (X = A * L^-T; pivot rows: D / undivided L*D)
            const int n = g & 3;
            const double xkn = n == 0 ? xk[0] : n == 1 ? xk[1] : n == 2 ? xk[2] : xk[3];
            if (below)
                akk[jp] = xkn;
            else if (r == c0 + 1 && n == 1)
                akk[jp] = f.d1;
            else if (r == c0 + 2 && n >= 1)
                akk[jp] = n == 1 ? f.u21 : f.d2;
            else if (r == c0 + 3 && n >= 1)
                akk[jp] = n == 1 ? f.u31 : n == 2 ? f.u32 : f.d3;
            ai[jp] = n == 0 ? xi[0] : n == 1 ? xi[1] : n == 2 ? xi[2] : xi[3];
            aj[jp] = n == 0 ? xj[0] : n == 1 ? xj[1] : n == 2 ? xj[2] : xj[3];
        }
        if (tid == 0) {
            s.z[c0] = z[0], s.z[c0 + 1] = z[1], s.z[c0 + 2] = z[2], s.z[c0 + 3] = z[3];
            s.iD[c0] = f.i0, s.iD[c0 + 1] = f.i1, s.iD[c0 + 2] = f.i2, s.iD[c0 + 3] = f.i3;
            s.Dv[c0] = f.d0, s.Dv[c0 + 1] = f.d1, s.Dv[c0 + 2] = f.d2, s.Dv[c0 + 3] = f.d3;
        }
        if (has_next) cur = nxt;
    }
    __syncthreads();   // the bookkeeping of the last iteration (s.z / s.iD / s.Dv) is read by the closing phase
#ifdef K7_TIMING
    const long long ts1 = (long long)__builtin_readcyclecounter();
#endif
    // ---- closing phase ----
    double id4[CPT];
#pragma unroll
    for (int jj = 0; jj < CPT; jj++) id4[jj] = s.iD[g + STRIPS * jj];
    if (role == 0) {
        // diagonal block of the factor buffer: Lkk^-T (upper triangular, unit diagonal) for the backward pass
#pragma unroll
        for (int jj = 0; jj < CPT; jj++) d.L[se_blk(k, k, band) + r * NB + g + STRIPS * jj] = ai[jj];
        if (tid < NB) {
            d.Dg[k * NB + tid] = s.Dv[tid];
            d.y[k * NB + tid] = s.z[tid];
        }
    } else if (role == 1) {
        // L_ik = X_i D^-1 ;  b_i -= L_ik z_k : per-thread partial, then the 8 threads of a row (consecutive lanes)
        double sum = 0;
#pragma unroll
        for (int jj = 0; jj < CPT; jj++) {
            const int q = g + STRIPS * jj;
            const double l = ai[jj] * id4[jj];
            d.L[se_blk(bi, k, band) + r * NB + q] = l;
            sum += l * s.z[q];
        }
        sum += dpp_row_shr_f64<1>(sum);   // the 8 threads of a row are 8 consecutive lanes of a DPP row: lane 7 collects
        sum += dpp_row_shr_f64<2>(sum);
        sum += dpp_row_shr_f64<4>(sum);
        static_assert(STRIPS == 8, "row sums below collect 8 lanes");
        if (g == STRIPS - 1) E[bi * NB + r] = e_old - sum;
    } else {
        // A_ij -= X_i D^-1 X_j^T  through LDS tiles
#pragma unroll
        for (int jj = 0; jj < CPT; jj++) {
            const int q = g + STRIPS * jj;
            s.Xi[r * LDP + q] = ai[jj];
            s.Xj[r * LDP + q] = (two ? aj[jj] : ai[jj]) * id4[jj];
        }
        __syncthreads();
        // 32x32x32 product on the matrix cores: each wave chains
        // eight v_mfma_f64_16x16x4_f64 (A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15];
        // D: column lane&15, row (lane>>4) + 4*v).  The vector version of this product was latency bound
        // (five LDS reads per four FMAs, one wave per SIMD) and took as long as three chain iterations.
        typedef double v4f64 __attribute__((ext_vector_type(4)));
        const int lane = mlane;
        const double* pa = s.Xi + (16 * qi + (lane & 15)) * LDP + (lane >> 4);
        const double* pbq = s.Xj + (16 * qj + (lane & 15)) * LDP + (lane >> 4);
        v4f64 acc = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < NB / 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk], pbq[4 * kk], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int row = 16 * qi + (lane >> 4) + 4 * v, col = 16 * qj + (lane & 15);
            S[se_blk(bi, bj, band) + row * NB + col] = sij[v] - acc[v];
        }
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

// w = D^-1 z ; L^T x = w, blocked backwards and right-looking.  One workgroup of 1024 threads:
// thread (rr, c) = (tid / 32, tid % 32).  Per block k:  x_k = Lkk^-T (w_k - pending_k)  is a 32x32
// mat-vec (row rr, 32-lane reduction), then  pending_j += L[k-block rows][j] . x_k  for every column j
// of the blocks above, four row-slices per column.
__global__ void __launch_bounds__(1024) ldlt_backward_kernel(BaDev d, int cur) {
    TL_MARK(d, 10)
    extern __shared__ __attribute__((aligned(16))) double bw_lds[];
    const int npad = d.npad, nblk = npad / NB, band = se_band(d);
    double* xs = bw_lds;         // npad: the solution
    double* pend = xs + npad;    // 4 x npad: sum_{blocks below} L^T x, one plane per row slice (summed in fixed order by the
                                 // reader: every rank of a sharded bundle must arrive at bit-identical poses — LDS atomics did not)
    double* wv = pend + 4 * npad;    // npad: D^-1 z
    const int tid = threadIdx.x;
    const int c = tid % NB, rr = tid / NB;
    for (int i = tid; i < npad; i += 1024) {
        pend[i] = pend[npad + i] = pend[2 * npad + i] = pend[3 * npad + i] = 0.0;
        wv[i] = d.y[i] / d.Dg[i];
    }
    double wnext = d.L[se_blk(nblk - 1, nblk - 1, band) + rr * NB + c];   // Lkk^-T element (rr, c)
    for (int k = nblk - 1; k >= 0; k--) {
        const double w = wnext;
        if (k > 0) wnext = d.L[se_blk(k - 1, k - 1, band) + rr * NB + c];   // prefetch the next block
        // first round of this block's update operands (independent of x_k): in flight during the mat-vec
        // columns of the blocks above that row block k reaches: all of them, or — S banded — the last `band` blocks
        const int jlo = max(0, k - band) * NB, ncol = k * NB - jlo;
        // block row k of L: its blocks left of the diagonal are contiguous, column j of row q sits at
        const double* Lrow = d.L + se_blk(k, max(0, k - band), band);
        auto l_at = [&](int q, int j) { return Lrow[(size_t)((j - jlo) / NB) * (NB * NB) + q * NB + ((j - jlo) % NB)]; };
        const bool upd = tid < ncol * 4;
        const int j0 = jlo + (upd ? tid % ncol : 0), sl0 = upd ? tid / ncol : 0;
        double lreg[8];
#pragma unroll
        for (int q = 0; q < 8; q++) lreg[q] = upd ? l_at(sl0 * 8 + q, j0) : 0.0;
        __syncthreads();   // (A) pending sums of the previous block are complete
        const int kc = k * NB + c;
        double p = (c >= rr) ? w * (wv[kc] - ((pend[kc] + pend[npad + kc]) + (pend[2 * npad + kc] + pend[3 * npad + kc]))) : 0.0;   // upper triangular
        // 32-lane sum on the VALU (DPP row shifts, then lane 15 of rows 0 / 2 into rows 1 / 3): five ds_bpermute
        // round trips sat on the block's critical path
        p += dpp_row_shr_f64<1>(p);
        p += dpp_row_shr_f64<2>(p);
        p += dpp_row_shr_f64<4>(p);
        p += dpp_row_shr_f64<8>(p);
        p += dpp_bcast_f64<0x142, 0xA>(p);
        if (c == NB - 1) xs[k * NB + rr] = p;
        __syncthreads();   // (B) x_k visible
        // pending sums of the blocks above: column j, rows of block k in four slices of 8
        if (upd) {
            double a = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) a += lreg[q] * xs[k * NB + sl0 * 8 + q];
            pend[sl0 * npad + j0] += a;   // (thread (slice, column) is the only writer of its slot in this round)
        }
        for (int idx = tid + 1024; idx < ncol * 4; idx += 1024) {
            const int j = jlo + idx % ncol, sl = idx / ncol;
            double a = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) a += l_at(sl * 8 + q, j) * xs[k * NB + sl * 8 + q];
            pend[sl * npad + j] += a;
        }
    }
    __syncthreads();
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

int ba_solve_init() {
    HIP_TRY(hipFuncSetAttribute((const void*)ldlt_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return PTAM_OK;
}

int ba_solve(ptam_ctx* ctx, BaDev& d, int cur) {
    const int nblk = d.npad / NB;
    for (int k = 0; k < nblk; k++) {
        const int rem = std::min(nblk - k - 1, se_band(d));
        const int nwg = 1 + rem + rem * (rem + 1) / 2;
        hipLaunchKernelGGL(ldlt_step_kernel, dim3(nwg), dim3(TPB), 0, ctx->stream, d, k);
    }
    const size_t bw_bytes = (size_t)6 * d.npad * sizeof(double);
    hipLaunchKernelGGL(ldlt_backward_kernel, dim3(1), dim3(1024), bw_bytes, ctx->stream, d, cur);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

// every kernel of this file resolved once, when a context is created: the first launch of a kernel otherwise pays for
// loading the code object / resolving the function — 10-28 ms in the middle of the first frame or the first adjustment
void solve_preload_kernels() {
    ptam_preload((const void*)ldlt_step_kernel);
    ptam_preload((const void*)ldlt_backward_kernel);
}
