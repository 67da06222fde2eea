// solve.hip — K9: dense SPD camera system  S * da = E  (src/Bundle.cc:457-458,
// `Cholesky<>(mS).backsub(vE)`: TooN's unpivoted LDL^T reading the lower triangle).
//
// Blocked LDL^T, block size 32, in the block band of S.  Which form runs (ba_solve, bottom of this file):
//   1 - 2 block rows                 ldlt_small_kernel: everything in one workgroup (ldlt_small.inc);
//   3 .. 13 dense, <= 28 banded      ONE persistent launch: a chain workgroup through the pivot blocks, a workgroup per block
//                                    row, one for the right-hand side, hand-offs by flags on one XCD (ldlt_chain.inc);
//   banded, nblk >= 2 band + 8       eliminated from BOTH ends: two such chains in one launch (the upward one on a mirrored
//                                    copy of the bottom end, on a second XCD), then the middle part as a third;
//   anything else (and A/B runs)     one launch per block column — the form described next (ldlt_step_body.inc; for a
//                                    two-ended elimination one launch per step of both chains, ldlt_step_twin_kernel);
// then ldlt_backward_kernel: D^-1 and the backward substitution, trial poses, |da|^2.
//
// The launch-per-block-column form: right-looking, ONE launch per block column k.  Every workgroup of
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
#define BW_LDS_MAX ((size_t)150 * 1024)   // dynamic LDS the backward substitution may take for its six vectors
static_assert(NB == 32, "the strip decomposition below is written for 32x32 blocks");

// reciprocal of a pivot: v_rcp_f64 (2^-24.4 relative, tools/pipes/rcpacc) and ONE cubic step y (1 + e + e^2), e = 1 - d y —
// three dependent FMAs instead of the four of two Newton steps, <= 1 ulp all the same; it sits on the chain of every pivot
__device__ __forceinline__ double fast_rcp(double d) {
    const double y = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, y, 1.0);
    return fma(y, fma(e, e, e), y);
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

typedef double v4f64 __attribute__((ext_vector_type(4)));
struct StepLds {
    double P[2][3][NB][4];   // the iteration's panel (raw strips of A_kk, A_i, A_j), double buffered
    double Pb[2][4];         // rhs entries of the pivot rows
    double Xi[NB * LDP];     // closing phase: X_i (and X_j D^-1) tiles
    double Xj[NB * LDP];
    double iD[NB], Dv[NB], z[NB];
};

// TWO-ENDED ("twisted") elimination of a banded system.  A banded SPD matrix can be eliminated from both ends at once: the
// block columns 0, 1, 2 ... downwards and nblk-1, nblk-2 ... upwards are independent chains as long as the tiles they update
// stay apart (a middle part of at least 2 * band blocks is left, then eliminated downwards as usual) — an elimination order
// like any other for an SPD system, only that it has two sequential chains of half the length.  One launch performs one step of
// EACH chain: the first `nwg_top` workgroups the downward step k, the others the upward step k_bot (k_bot < 0: none).
// The upward step is the same code on mirrored coordinates: within its pivot block rows / columns run backwards
// (element (r, q) is A(31-q, 31-r) of the stored lower triangle), its panel blocks A(bi, k_bot), bi < k_bot, are the
// transposes of the stored blocks (k_bot, bi); everything it leaves behind for the substitution (Lkk^-T, D, z, the
// multiplier blocks) is expressed in the mirrored coordinates of block k_bot.
// row_limit: blocks >= row_limit do not exist for the downward step (they were eliminated by the other chain).
__global__ void __launch_bounds__(TPB) ldlt_step_kernel(BaDev d, int k, int row_limit) {
    if (k == 0) { TL_MARK(d, 9) }
    __shared__ StepLds s;
    constexpr bool mir = false;
    const int wg = blockIdx.x;
    const int rem = min(row_limit - k - 1, se_band(d));   // (outside the band column k holds zeros: not stored)
#include "ldlt_step_body.inc"
}

// one step of each chain of a two-ended elimination: the first nwg_top workgroups the downward step k_top, the others the
// upward step k_bot
__global__ void __launch_bounds__(TPB) ldlt_step_twin_kernel(BaDev d, int k_top, int k_bot, int nwg_top) {
    if (k_top == 0) { TL_MARK(d, 9) }
    __shared__ StepLds s;
    if ((int)blockIdx.x >= nwg_top) {
        constexpr bool mir = true;
        const int k = k_bot, wg = (int)blockIdx.x - nwg_top;
        const int rem = min(k, se_band(d));
#include "ldlt_step_body.inc"
    } else {
        constexpr bool mir = false;
        const int k = k_top, wg = blockIdx.x;
        const int rem = min(d.npad / NB - k - 1, se_band(d));
#include "ldlt_step_body.inc"
    }
}

#include "ldlt_small.inc"
#include "ldlt_chain.inc"

// w = D^-1 z ; L^T x = w, blocked backwards and right-looking.  One workgroup of 1024 threads:
// thread (rr, c) = (tid / 32, tid % 32).  Per block k:  x_k = Lkk^-T (w_k - pending_k)  is a 32x32
// mat-vec (row rr, 32-lane reduction), then  pending_j += L[k-block rows][j] . x_k  for every column j
// of the blocks above, four row-slices per column.
// b_start: first block of the upward chain of a two-ended elimination (nblk: none).  The substitution runs in reverse
// elimination order: the middle and the downward chain's blocks from b_start - 1 down to 0 (right-looking, as always), then
// the upward chain's blocks from b_start up to the end — block k of that chain couples to the `band` blocks above it, whose
// solution is known by then (left-looking: a (band * 32)-column mat-vec with the stored multiplier blocks, then the 32x32
// triangular one), in the chain's mirrored coordinates.
// TWO workgroups when the elimination was two-ended: the chains are independent given the middle's solution, so workgroup 0
// substitutes the middle and the downward chain's blocks, workgroup 1 the middle AGAIN (2 * band blocks, bit-identical) and
// then the upward chain's blocks — 22 sequential block steps each instead of 38 in a row at 200 keyframes / window 16.
// Each writes its part of da, the trial poses of the cameras whose rows it knows, and its part of |da|^2 (d.sumsq2).
template <bool BIG>
__global__ void __launch_bounds__(1024) ldlt_backward_kernel(BaDev d, int cur, int b_start) {
    TL_MARK(d, 10)
    const int chain = blockIdx.x;   // 0: middle + downward chain (or everything, one-ended); 1: middle + upward chain
    extern __shared__ __attribute__((aligned(16))) double bw_lds[];
    const int npad = d.npad, nblk = npad / NB, band = se_band(d);
    // (systems of more than ~560 cameras: the six vectors do not fit LDS and live in global memory — one workgroup reads
    //  what it wrote itself, behind its own barriers)
    double* xs = BIG ? d.bw_scratch + (size_t)chain * 6 * npad : bw_lds;   // npad: the solution
    double* pend = xs + npad;    // 4 x npad: sum_{blocks below} L^T x, one plane per row slice (summed in fixed order by the
                                 // reader: every rank of a sharded bundle must arrive at bit-identical poses — LDS atomics did not)
    double* wv = pend + 4 * npad;    // npad: D^-1 z
    const int tid = threadIdx.x;
    const int c = tid % NB, rr = tid / NB;
    // (the persistent factorisation's error word: a spin gave up during THIS solve — the host stops the adjustment)
    // (the persistent factorisation's error word: a wait gave up during one of THIS solve's launches — up to three, with three
    //  sequence numbers.  The word is taken down again: the host repeats the trial with the launch-per-block-column form)
    if (tid == 0 && chain == 0 && d.sflags && d.sflags[0] != 0) {
        d.sc->solve_fault = 1;
        d.sflags[0] = 0;
    }
    for (int i = tid; i < npad; i += 1024) {
        pend[i] = pend[npad + i] = pend[2 * npad + i] = pend[3 * npad + i] = 0.0;
        wv[i] = d.y[i] / d.Dg[i];
    }
    const int t_end = nblk - b_start;                 // blocks of each chain (0: one-ended elimination)
    const int k_stop = chain == 0 ? 0 : t_end;        // workgroup 1 stops above the middle
    double wnext = d.L[se_blk(b_start - 1, b_start - 1, band) + rr * NB + c];   // Lkk^-T element (rr, c)
#ifdef K7_TIMING
#define BW_STAMP(i) if (tid == 0 && chain == 0) d.dbg[1000 + 8 * k + (i)] = (long long)__builtin_readcyclecounter();
#else
#define BW_STAMP(i)
#endif
    // work item of the updates = (tile, slice of 8 rows, column): thread -> (tid >> 7, (tid >> 5) & 3, tid & 31), eight tiles of
    // L(k, .) per round.  (No division by a run-time count anywhere: `tid % ncol`, `tid / ncol` — two integer divisions per
    // thread and block — were 400 of a block's 3 300 cycles, on the chain.)  What stands between two blocks now is the
    // requests themselves: 16 waves x 9 loads x 512 B are ~1 000 cycles of the CU's one memory pipeline at 10 block rows, and
    // the waves wait for queue space wherever the requests are placed (issued during the previous block's mat-vec they made
    // THAT 900 cycles longer: measured, no gain).
    const int ucol = tid & (NB - 1), sl0 = (tid >> 5) & 3, ut0 = tid >> 7;
    auto fetch_l = [&](int k, double (&l)[8]) {
        const int jb0 = max(0, k - band);
        const double* src = d.L + se_blk(k, jb0, band) + (size_t)ut0 * (NB * NB) + (sl0 * 8) * NB + ucol;
        const bool on = ut0 < k - jb0;
#pragma unroll
        for (int q = 0; q < 8; q++) l[q] = on ? src[q * NB] : 0.0;
    };
    for (int k = b_start - 1; k >= k_stop; k--) {
        const double w = wnext;
        if (k > k_stop) wnext = d.L[se_blk(k - 1, k - 1, band) + rr * NB + c];   // prefetch the next block
        double lreg[8];
        fetch_l(k, lreg);   // this block's update operands (independent of x_k): in flight during the mat-vec
        // columns of the blocks above that row block k reaches: all of them, or — S banded — the last `band` blocks
        const int jb0 = max(0, k - band), nbk = k - jb0, jlo = jb0 * NB;   // nbk tiles L(k, jb0 .. k-1), contiguous in block row k of L
        const double* Lrow = d.L + se_blk(k, jb0, band);
        const bool upd = ut0 < nbk;
        const int j0 = jlo + ut0 * NB + ucol;
        BW_STAMP(0)
        __syncthreads();   // (A) pending sums of the previous block are complete
        BW_STAMP(1)
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
        BW_STAMP(2)
        __syncthreads();   // (B) x_k visible
        BW_STAMP(3)
        // pending sums of the blocks above: column j, rows of block k in four slices of 8
        if (upd) {
            double a = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) a += lreg[q] * xs[k * NB + sl0 * 8 + q];
            pend[sl0 * npad + j0] += a;   // (thread (slice, column) is the only writer of its slot in this round)
        }
        for (int ut = ut0 + 8; ut < nbk; ut += 8) {   // (more than eight tiles in the block row)
            double a = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) a += Lrow[(size_t)ut * (NB * NB) + (sl0 * 8 + q) * NB + ucol] * xs[k * NB + sl0 * 8 + q];
            pend[sl0 * npad + jlo + ut * NB + ucol] += a;
        }
        BW_STAMP(4)
    }
    __syncthreads();
    for (int k = b_start; k < nblk && chain == 1; k++) {
        const int jb = max(0, k - band), nj = k - jb;
        const double w = d.L[se_blk(k, k, band) + rr * NB + c];   // Lkk^-T element (rr, c), mirrored coordinates
        // thread (q = rr, lane c): row q of the stored blocks (k, jb .. k-1) against the known solution of those blocks
        const double* Lrow = d.L + se_blk(k, jb, band) + rr * NB + c;
        double a = 0;
        for (int j = 0; j < nj; j++) a += Lrow[(size_t)j * (NB * NB)] * xs[(jb + j) * NB + c];
        a += dpp_row_shr_f64<1>(a);
        a += dpp_row_shr_f64<2>(a);
        a += dpp_row_shr_f64<4>(a);
        a += dpp_row_shr_f64<8>(a);
        a += dpp_bcast_f64<0x142, 0xA>(a);
        if (c == NB - 1) pend[k * NB + rr] = a;   // (the planes of `pend` past b_start are free: plane 0 holds the sums)
        __syncthreads();
        const int kc = k * NB + c;
        double p = (c >= rr) ? w * (wv[kc] - pend[kc]) : 0.0;   // upper triangular
        p += dpp_row_shr_f64<1>(p);
        p += dpp_row_shr_f64<2>(p);
        p += dpp_row_shr_f64<4>(p);
        p += dpp_row_shr_f64<8>(p);
        p += dpp_bcast_f64<0x142, 0xA>(p);
        if (c == NB - 1) xs[k * NB + (NB - 1 - rr)] = p;   // back to natural coordinates
        __syncthreads();
    }
    // my part of the solution: rows [r_lo, r_hi)
    const bool two = t_end > 0;
    const int r_lo = chain == 0 ? 0 : b_start * NB, r_hi = (two && chain == 0) ? b_start * NB : npad;
    for (int i = r_lo + tid; i < r_hi; i += 1024) d.da[i] = xs[i];
    // trial poses  exp(da_j) * se3CfW  (src/Bundle.cc:496-501) and |da|^2, straight from LDS: saves the
    // separate pose-update launch.  A camera whose six rows reach into the upward chain's blocks belongs to workgroup 1
    // (which knows the middle as well), every other one — the fixed ones included — to workgroup 0.
    for (int c2 = tid; c2 < d.C; c2 += 1024) {
        const int f = d.cam_free[c2];
        const bool mine = (two && f >= 0 && 6 * f + 5 >= b_start * NB) ? chain == 1 : chain == 0;
        if (!mine) continue;
        const double* T = d.pose[cur] + 12 * c2;
        double* Tn = d.pose[cur ^ 1] + 12 * c2;
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
        for (int i = r_lo + tid; i < min(r_hi, d.n); i += 64) sq += xs[i] * xs[i];
        sq = wave_sum_f64(sq);
        if (tid == 0) {
            d.sumsq2[chain] = sq;   // (finalize_new_kernel adds the two)
            if (!two) d.sumsq2[1] = 0.0;
        }
    }
}

int ba_solve_init() {
    HIP_TRY(hipFuncSetAttribute((const void*)ldlt_backward_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)ldlt_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)ldlt_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return PTAM_OK;
}

// blocks the downward chain eliminates before the middle part of a two-ended elimination (0: plain top-down elimination).
// The two chains must never update the same tile: the middle keeps at least 2 * band blocks.
static int ldlt_twist_len(int nblk, int band) {
    static const bool off = ptam_ab_env("PTAM_LDLT_ONE_ENDED") != nullptr;   // (A/B runs)
    if (off || band < 1 || nblk < 2 * band + 8) return 0;   // (at least four block columns per chain: below that the mirrored copy and the third launch cost what the shorter chain saves)
    return (nblk - 2 * band) / 2;
}

size_t ba_solve_flag_bytes(int nblk) { return ch_flag_words(nblk, nblk) * sizeof(unsigned); }

static ChainArgs chain_args_natural(const BaDev& d, int k0, int k1, int kr) {
    return ChainArgs{d.SE, se_E(d), d.L, d.Dg, d.y, d.sflags, k0, k1, kr, d.n};
}

int ba_solve(ptam_ctx* ctx, BaDev& d, int cur) {
    const int nblk = d.npad / NB, band = se_band(d);
    static const bool no_small = ptam_ab_env("PTAM_LDLT_NO_SMALL") != nullptr;   // (A/B runs: launch-per-block-column form only)
    // one or two block rows in one workgroup and one launch: ldlt_small.inc.  (It holds up to SM_NB = 5 block rows and was the
    // form of every system up to that size until the persistent launch's row workers learnt to keep up with the chain; now,
    // us per solve, small / persistent: 10.3 / 10.7 at 1 block row, 16.8 / 16.5 at 2, 25.7 / 23.8 at 3, 36.1 / 30.6 at 4,
    // 48.7 / 37.7 at 5 — the single workgroup does the row updates one after the other.)
    if (!no_small && nblk <= SM_USE_NB) {
        hipLaunchKernelGGL(ldlt_small_kernel, dim3(1), dim3(TPB), sizeof(SmallLds), ctx->stream, d, cur);
        HIP_TRY(hipGetLastError());
        return PTAM_OK;
    }
    auto nwg_of = [](int rem) { return 1 + rem + rem * (rem + 1) / 2; };
    const int t_end = ldlt_twist_len(nblk, band), b_start = nblk - t_end;
    {
        // not banded enough for two chains (below): one persistent launch (ldlt_chain.inc) when a row worker's in-band tiles fit its
        // LDS and the block rows — a workgroup each — fit one XCD: measured against the
        // launch-per-block-column form below (tools/ldlt, us per solve) 49 / 67 at 7 block rows, 71 / 98 at 10, 86 / 119 at 12,
        // 96 / 133 at 13 (dense: from 14 block rows on a worker's tiles no longer fit), 105 / 149 at 15 rows of band 5,
        // 202 / 287 at 28 of band 9.  (Where two chains apply they win: 137 against 152 at 23 rows of band 2.)
        static const bool no_chain = getenv("PTAM_LDLT_NO_CHAIN") != nullptr;   // (A/B runs)
        // (the right-hand-side workgroup of that launch also substitutes backwards — ldlt_chain.inc, "and backwards" — when
        //  its vectors fit beside a row worker's tiles; PTAM_LDLT_SEPARATE_BACKWARD=1: ldlt_backward_kernel behind it, A/B runs)
        static const bool sep_bw = getenv("PTAM_LDLT_SEPARATE_BACKWARD") != nullptr;
        // (measured, tools/ldlt, us per solve inside / behind the launch: 28.8 / 29.7 at 4 block rows, 47.4 / 47.9 at 7, 69.0 / 68.4 at 10 — one
        //  compute unit fetches a tile from the L2 in ~200 cycles, 34 B per cycle, whoever asks; so only where the rows are short)
        const bool bw_in = !sep_bw && band <= CH_BW_MAXT && ch_lds_bytes_bw(band, nblk) <= CH_LDS_MAX;
        const size_t lds = bw_in ? ch_lds_bytes_bw(band, nblk) : ch_lds_bytes(band);
        if (t_end == 0 && !no_chain && !d.chain_off && d.sflags && nblk <= CH_MAX_NB && lds <= CH_LDS_MAX) {
            d.solve_seq++;
            if (d.solve_seq >= (1u << 27)) d.solve_seq = 1;   // (flags carry it shifted by up to 4 bits; flags of 2^27 solves ago are no concern)
            {
                const ChainArgs a = chain_args_natural(d, 0, nblk, nblk);
                hipLaunchKernelGGL(ldlt_chain_kernel, dim3(8 * ch_roles(nblk)), dim3(TPB), lds, ctx->stream, d, a, a, 1, cur, bw_in ? 1 : 0);
            }
            if (!bw_in) {
                const size_t bw = (size_t)6 * d.npad * sizeof(double);
                hipLaunchKernelGGL(ldlt_backward_kernel<false>, dim3(1), dim3(1024), bw, ctx->stream, d, cur, nblk);
            }
            HIP_TRY(hipGetLastError());
            return PTAM_OK;
        }
    }
    // Two chains: as TWO persistent chains of one launch — the downward one on S itself (blocks 0, 8, 16 ... of the launch: XCD
    // 0), the upward one on a mirrored copy of the system's bottom end (blocks 1, 9, 17 ...: XCD 1; ldlt_chain.inc, "the bottom
    // end") — where each chain's block rows fit one XCD; as one launch per step of both chains otherwise.
    static const bool no_chain2 = getenv("PTAM_LDLT_NO_CHAIN") != nullptr || ptam_ab_env("PTAM_LDLT_TWIN_LAUNCHES") != nullptr;   // (A/B runs)
    const int kr2 = std::min(b_start, t_end + band);   // (each chain's rows: its columns and the `band` block rows they reach)
    const bool two_persistent = t_end >= 2 && !no_chain2 && !d.chain_off && d.sflags && d.SE2 && kr2 <= CH_MAX_NB && ch_lds_bytes(band) <= CH_LDS_MAX;
    if (two_persistent) {
        d.solve_seq++;
        if (d.solve_seq >= (1u << 27)) d.solve_seq = 1;
        hipLaunchKernelGGL(ldlt_mirror_in_kernel, dim3(kr2 * (band + 1)), dim3(TPB), 0, ctx->stream, d, kr2);
        const ChainArgs a0 = chain_args_natural(d, 0, t_end, kr2);
        const ChainArgs a1{d.SE2, d.SE2 + se_size(nblk, band), d.L2, d.Dg2, d.y2, d.sflags2, 0, t_end, kr2, d.npad};   // (mirrored: the padding comes first, nothing to skip)
        hipLaunchKernelGGL(ldlt_chain_kernel, dim3(8 * ch_roles(kr2)), dim3(TPB), ch_lds_bytes(band), ctx->stream, d, a0, a1, 2, cur, 0);
        hipLaunchKernelGGL(ldlt_mirror_out_kernel, dim3(kr2 * (band + 1)), dim3(TPB), 0, ctx->stream, d, t_end, kr2);
    }
    for (int st = 0; st < (two_persistent ? 0 : t_end); st++) {   // one step of each chain per launch
        const int kt = st, kb = nblk - 1 - st;
        const int nt = nwg_of(std::min(nblk - kt - 1, band)), nb = nwg_of(std::min(kb, band));
        hipLaunchKernelGGL(ldlt_step_twin_kernel, dim3(nt + nb), dim3(TPB), 0, ctx->stream, d, kt, kb, nt);
    }
    {
        // the middle (everything, without a second chain): one persistent launch over its block rows where that form applies
        // (ldlt_chain.inc: 5.6 us per block row against 8.3 per launch), one launch per block column otherwise
        static const bool no_chain = getenv("PTAM_LDLT_NO_CHAIN") != nullptr;
        const int n_mid = b_start - t_end;
        const size_t lds = ch_lds_bytes(band);
        if (t_end > 0 && !no_chain && !d.chain_off && d.sflags && n_mid >= 3 && n_mid <= CH_MAX_NB && lds <= CH_LDS_MAX) {
            d.solve_seq++;
            if (d.solve_seq >= (1u << 27)) d.solve_seq = 1;
            {
                const ChainArgs a = chain_args_natural(d, t_end, b_start, b_start);
                hipLaunchKernelGGL(ldlt_chain_kernel, dim3(8 * ch_roles(n_mid)), dim3(TPB), lds, ctx->stream, d, a, a, 1, cur, 0);
            }
        } else {
            for (int k = t_end; k < b_start; k++) {
                const int nwg = nwg_of(std::min(b_start - k - 1, band));
                hipLaunchKernelGGL(ldlt_step_kernel, dim3(nwg), dim3(TPB), 0, ctx->stream, d, k, b_start);
            }
        }
    }
    const size_t bw_bytes = (size_t)6 * d.npad * sizeof(double);
    if (bw_bytes > BW_LDS_MAX)   // (the vectors in d.bw_scratch instead)
        hipLaunchKernelGGL(ldlt_backward_kernel<true>, dim3(t_end > 0 ? 2 : 1), dim3(1024), 0, ctx->stream, d, cur, b_start);
    else
        hipLaunchKernelGGL(ldlt_backward_kernel<false>, dim3(t_end > 0 ? 2 : 1), dim3(1024), bw_bytes, ctx->stream, d, cur, b_start);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

// every kernel of this file resolved once, when a context is created: the first launch of a kernel otherwise pays for
// loading the code object / resolving the function — 10-28 ms in the middle of the first frame or the first adjustment
void solve_preload_kernels() {
    ptam_preload((const void*)ldlt_step_kernel);
    ptam_preload((const void*)ldlt_step_twin_kernel);
    ptam_preload((const void*)ldlt_backward_kernel<false>);
    ptam_preload((const void*)ldlt_backward_kernel<true>);
    ptam_preload((const void*)ldlt_small_kernel);
    ptam_preload((const void*)ldlt_chain_kernel);
    ptam_preload((const void*)ldlt_mirror_in_kernel);
    ptam_preload((const void*)ldlt_mirror_out_kernel);
}
