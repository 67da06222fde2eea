// bundle.hip — Bundle::Compute (src/Bundle.cc:116-158) and Do_LM_Step (:209-551) on gfx950.
//
// Kernels (DESIGN.md has the byte counts):
//   K5  project_e2_kernel     pass 1 (:219-225, :164-180) + first-level histogram of e^2
//   K6  select_*_kernel       exact order statistic sorted[n/2] (include/Tools.h:152-162) by radix
//                             select on the fp64 bit pattern -> sigma^2 (floor MinTukeySigma^2 :234-237)
//   K7  jac_accum_kernel      pass 2 (:250-332) FUSED: weight, A(2x6), B(2x3), U += A^T A, epsA,
//                             V += B^T B, epsB, W = A^T B.  Roofline kernel (HBM-bound, ~185 B/meas).
//       reduce_partials_kernel  fixed-order sum of the per-workgroup camera partials
//   K8a vinv_kernel           V*^-1 (:341-359)
//   K8  schur_tile_mfma_kernel S = U* - sum_i (W V*^-1) W^T, E = epsA - sum_i W V*^-1 epsB (:374-446),
//                             output-stationary over 8x8-camera tiles; schur_reduce_kernel sums the
//                             tile partials in fixed order (deterministic) and writes S lower / E
//   K9  solve.hip             blocked LDL^T + substitutions (:457-458)
//   K10 pose_update_kernel, point_update_kernel, finalize_new_kernel
//                             delta b (:461-483), update norm (:488-490), trial poses/points
//                             (:496-504), new robust error (:188-207, :506)
//       purge_kernel          erase bad measurements, record outliers (:536-547)
// The lambda-trial control flow, convergence test and abort polling stay on the host, one readback of
// a 64-byte scalar block per trial.
#include <algorithm>
#include <chrono>
#include <climits>
#include <atomic>
#include <numeric>
#include <utility>

#include "bundle.h"

// =================================================================================================
// device helpers
// =================================================================================================
// First-level bin of the order-statistic select: the top 16 bits of the fp64 pattern (exponent + 4
// mantissa bits = 16 sub-bins per binade) relative to 2^-24, clamped to [0, 4095] — monotone in the
// value, 256 binades wide (6e-8 .. 7e69).  A rank that lands in a clamped end bin takes the generic
// full-width radix path of select_final_kernel.
#define E2_BIN_BASE ((1023 - 24) << 4)
__device__ __forceinline__ int e2_bin(double e2) {
    const int t = (int)((unsigned long long)__double_as_longlong(e2) >> 48) - E2_BIN_BASE;
    return t < 0 ? 0 : (t > HIST_BINS - 1 ? HIST_BINS - 1 : t);
}
__device__ __forceinline__ int e2_bin2(double e2) {   // next 12 bits
    return (int)(((unsigned long long)__double_as_longlong(e2) >> 36) & (HIST_BINS - 1));
}

// Kernels of the NEXT LM step are enqueued before the host has seen the trial's outcome (the GPU would idle ~10 us
// per trial otherwise).  They carry d.guard != 0 and leave at once unless finalize_new_kernel decided on the device
// what the host is about to decide from the same numbers.
__device__ __forceinline__ bool ba_guard_blocks(const BaDev& d) {
    if (d.guard == 0) return false;
    return d.guard == 1 ? (d.sc->spec_go == 0 && d.sc->spec_stay == 0) : d.sc->end_step == 0;
}

// project one measurement: returns false (bad) if z <= 0  (ProjectAndFindSquaredError :164-180).
// Same arithmetic as cam_project / cam_derivs (common.h) with the divisions replaced by three
// Newton-refined reciprocals (1/z, 1/r, 1/(1+k^2 r^2)); pass 1, pass 2 and the new-error pass all
// go through this one function, so they see bit-identical projections.
struct BaProj {
    double X, Y, Z;      // v3Cam
    double x, y;         // z = 1 plane
    double u, v;         // image
    double r, ir, f;     // radius, 1/r, rtrans_factor
};
// atan for x >= 0 in ~45 VALU instructions (the library call is ~80 and sits in K7's critical VALU
// budget): argument reduction atan(x) = atan(c) + atan((x - c) / (1 + c x)) with c in {0, 1/2, 1, 3/2}
// (x >= 39/16: pi/2 - atan(1/x)) and the classic 11-term odd minimax polynomial on |t| < 7/16
// (Sun fdlibm's published coefficients); the quotient uses rcp + 2 Newton steps.  Measured against
// libm over [1e-8, 1e3]: <= 2 ulp (constants: tests/test_oracle_kat.py::test_atan_reduction_constants;
// the device code itself is covered by the BA parity tests).
// The constants come from a __constant__ table: uniform loads put them in SGPRs, which a VOP3 fp64
// instruction reads directly — as literals each use would cost two v_mov_b32 and a VGPR pair.
__constant__ double BA_ATAN_K[20] = {
    3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,  -1.11111104054623557880e-01,
    9.09088713343650656196e-02,  -7.69187620504482999495e-02, 6.66107313738753120669e-02,  -5.83357013379057348645e-02,
    4.97687799461593236017e-02,  -3.65315727442169155270e-02, 1.62858201153657823623e-02,
    4.63647609000806093515e-01,  7.85398163397448278999e-01,  9.82793723247329054082e-01,  1.57079632679489655800e+00,
    0.4375, 0.6875, 1.1875, 2.4375, 1.5};
__device__ __forceinline__ double ba_atan_pos(double x) {
    const double* __restrict__ K = BA_ATAN_K;
    double c = 0.0, hi = 0.0;
    if (x >= K[15]) c = 0.5, hi = K[11];
    if (x >= K[16]) c = 1.0, hi = K[12];
    if (x >= K[17]) c = K[19], hi = K[13];
    double num = x - c, den = fma(c, x, 1.0);
    if (x >= K[18]) {
        num = -1.0;
        den = x;
        hi = K[14];
    }
    const double t = num * rcp_nr(den);
    const double z = t * t, w = z * z;
    // v_fma_f64 with the coefficient as an SGPR operand (left alone the compiler picks v_fmac and
    // first copies every coefficient into a VGPR pair)
#define FMA_SK(a, b, k) ({ double r_; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r_) : "v"(a), "v"(b), "s"(k)); r_; })
    double p1 = FMA_SK(w, K[10], K[8]);   // (K[10] is copied once: two scalar operands are not encodable)
    p1 = FMA_SK(w, p1, K[6]);
    p1 = FMA_SK(w, p1, K[4]);
    p1 = FMA_SK(w, p1, K[2]);
    p1 = FMA_SK(w, p1, K[0]);
    double p2 = FMA_SK(w, K[9], K[7]);
    p2 = FMA_SK(w, p2, K[5]);
    p2 = FMA_SK(w, p2, K[3]);
    p2 = FMA_SK(w, p2, K[1]);
#undef FMA_SK
    const double s1 = z * p1, s2 = w * p2;
    return hi - (t * (s1 + s2) - t);
}
// 1/sqrt(a), a > 0: v_rsq_f64 + 2 Newton steps (<= 1 ulp)
__device__ __forceinline__ double rsq_nr(double a) {
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    return y;
}
__device__ __forceinline__ bool ba_project(const DevCam& cam, const double* __restrict__ T, const double* __restrict__ X,
                                           BaProj& p) {
    se3_apply(T, X[0], X[1], X[2], p.X, p.Y, p.Z);
    if (p.Z <= 0) return false;
    const double iz = rcp_nr(p.Z);
    p.x = p.X * iz;
    p.y = p.Y * iz;
    const double r2 = p.x * p.x + p.y * p.y;
    if (r2 < 1e-6 || cam.w == 0.0) {   // r < 0.001  (src/ATANCamera.h:105-111)
        p.r = r2 > 0 ? r2 * rsq_nr(r2) : 0.0;
        p.ir = 0;
        p.f = 1.0;
    } else {
        p.ir = rsq_nr(r2);
        p.r = r2 * p.ir;
        const double t = p.r * cam.two_tan;   // (negative only for a negative FOV parameter)
        p.f = cam.w_inv * copysign(ba_atan_pos(fabs(t)), t) * p.ir;
    }
    p.u = cam.cx + cam.fx * (p.f * p.x);
    p.v = cam.cy + cam.fy * (p.f * p.y);
    return true;
}
// GetProjectionDerivs (src/ATANCamera.cc:179-209): dFdx = x*g, dFdy = y*g with the common factor
// g = [ (k/w)/(1+k^2 r^2) - f ] / r^2
__device__ __forceinline__ void ba_derivs(const DevCam& cam, const BaProj& p, double D[4]) {
    const double r = p.r * cam.dist_enabled;
    double g = 0.0;
    if (!(r < 0.01)) {
        const double k = cam.two_tan;
        const double iden = rcp_nr(1 + k * k * r * r);
        g = (cam.w_inv * k * iden - p.f) * (p.ir * p.ir);
    }
    const double dx = p.x * g, dy = p.y * g;
    D[0] = cam.fx * (dx * p.x + p.f);
    D[2] = cam.fy * (dx * p.y);
    D[1] = cam.fx * (dy * p.x);
    D[3] = cam.fy * (dy * p.y + p.f);
}
// Tukey with a precomputed 1/sigma^2 (other estimators keep their general form)
__device__ __forceinline__ double ba_sqrt_weight(int est, double e2, double s2, double is2) {
    if (est == PTAM_EST_TUKEY) return e2 > s2 ? 0.0 : 1.0 - e2 * is2;
    return est_sqrt_weight(est, e2, s2);
}
__device__ __forceinline__ double ba_objective(int est, double e2, double s2, double is2) {
    if (est == PTAM_EST_TUKEY) {
        if (e2 > s2) return 1.0;
        const double dd = 1.0 - e2 * is2;
        return 1.0 - dd * dd * dd;
    }
    return est_objective(est, e2, s2);
}

// =================================================================================================
// K5: pass 1
// =================================================================================================
__global__ void __launch_bounds__(BA_CHUNK) project_e2_kernel(DevCam cam, BaDev d, int cur, int build_hist) {
    TL_MARK(d, 13)
    __shared__ unsigned hist[HIST_BINS];
    const int tid = threadIdx.x;
    if (build_hist)
        for (int b = tid; b < HIST_BINS; b += BA_CHUNK) hist[b] = 0;
    __syncthreads();
    const double* __restrict__ pose = d.pose[cur];
    const double* __restrict__ pt = d.pt[cur];
    for (int ci = blockIdx.x; ci < d.n_chunks; ci += gridDim.x) {
        const BaChunk ch = d.chunks[ci];
        const int m = ch.m_begin + tid;
        if (m < ch.m_end) {
            const int st = d.m_state[m];
            if (st != MS_DEAD) {
                const int c = d.m_cam[m], p = d.m_pt[m];
                BaProj pr;
                if (!ba_project(cam, pose + 12 * c, pt + 3 * p, pr)) {
                    d.m_state[m] = MS_BAD;
                } else {
                    const double2 fo = d.m_found[m];
                    const double s = d.m_s[m];
                    const double ex = s * (fo.x - pr.u), ey = s * (fo.y - pr.v);
                    const double e2 = ex * ex + ey * ey;
                    d.m_e2[m] = e2;
                    if (st != MS_ALIVE) d.m_state[m] = MS_ALIVE;
                    if (build_hist) atomicAdd(&hist[e2_bin(e2)], 1u);
                }
            }
        }
    }
    if (build_hist) {
        __syncthreads();
        for (int b = tid; b < HIST_BINS; b += BA_CHUNK) {
            const unsigned c = hist[b];
            if (c) atomicAdd(&d.hist[b], c);
        }
    }
}

// Pass 1 of the step that follows an ACCEPTED trial: the trial's new-error pass (point_update_kernel)
// already projected every measurement with what are now the current poses / points, so this only
// adopts its squared errors and z <= 0 flags and builds the histogram — no projection (K5 proper is project_e2_kernel above).  With PURGE it first closes the
// finished step (purge_kernel's body: erase bad measurements, append to the outlier list :536-547).
// P1_U measurements per thread, every load issued (clamped, unconditional) before the first use: the
// kernel is a chain of dependent round trips otherwise (state -> flag -> e^2), ~1 us each.
#ifndef P1_U
#define P1_U 8
#endif
template <bool PURGE>
__device__ __forceinline__ void pass1_adopt(const BaDev& d, bool adopt, bool build_hist, unsigned* hist, bool keep = false) {
    constexpr int U = P1_U;
    const int tid = threadIdx.x, lane = tid & 63;
    const int m_end = (d.M + 63) & ~63;   // whole waves stay together for the ballot
    for (int base = blockIdx.x * (256 * U); base < m_end; base += gridDim.x * (256 * U)) {
        int m[U], st[U], zb[U];
        double e2[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            m[u] = base + u * 256 + tid;
            const int mc = min(m[u], d.M - 1);
            st[u] = m[u] < d.M ? (int)d.m_state[mc] : (int)MS_DEAD;
            zb[u] = d.m_zbad_t[mc];
            e2[u] = keep ? d.m_e2[mc] : d.m_e2t[mc];   // (keep: the state did not move, pass 1's own errors stand — pass1_keep_kernel)
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool in = m[u] < d.M;
            if (PURGE) {
                const bool bad = st[u] == MS_BAD;
                const unsigned long long mask = __ballot(bad);
                if (mask != 0) {
                    int at = 0;
                    if (lane == 0) at = atomicAdd(&d.sc->n_outliers, __popcll(mask));   // one atomic per wave
                    at = __shfl(at, 0, 64);
                    if (bad) {
                        d.outliers[at + __popcll(mask & ((1ull << lane) - 1ull))] = d.m_orig[m[u]];
                        d.m_state[m[u]] = MS_DEAD;
                        st[u] = MS_DEAD;
                    }
                }
            }
            if (keep) {
                if (in && st[u] == MS_ALIVE && build_hist) atomicAdd(&hist[e2_bin(e2[u])], 1u);
                continue;
            }
            if (!adopt || !in) continue;
            if (st[u] == MS_DEAD) continue;
            if (zb[u]) {
                d.m_state[m[u]] = MS_BAD;
            } else {
                d.m_e2[m[u]] = e2[u];
                if (st[u] != MS_ALIVE) d.m_state[m[u]] = MS_ALIVE;
#ifndef P1_NOLDS
                if (build_hist) atomicAdd(&hist[e2_bin(e2[u])], 1u);
#endif
            }
        }
    }
}
__device__ __forceinline__ void hist_clear(unsigned* hist) {
    for (int b = threadIdx.x; b < HIST_BINS; b += 256) hist[b] = 0;
    __syncthreads();
}
__device__ __forceinline__ void hist_flush(const BaDev& d, const unsigned* hist) {
    __syncthreads();
    for (int b = threadIdx.x; b < HIST_BINS; b += 256) {
        const unsigned c = hist[b];
#ifndef P1_NOFLUSH
        if (c) atomicAdd(&d.hist[b], c);
#else
        if (c == 0xffffffffu) d.hist[b] = c;
#endif
    }
}
__global__ void __launch_bounds__(256) pass1_from_trial_kernel(BaDev d, int build_hist) {
    TL_MARK(d, 14)
    if (ba_guard_blocks(d)) return;
    __shared__ unsigned hist[HIST_BINS];
    if (build_hist) hist_clear(hist);
    pass1_adopt<false>(d, true, build_hist != 0, hist);
    if (build_hist) hist_flush(d, hist);
}
// Pass 1 of a step that follows a step WITHOUT an accepted trial (every trial rejected until the iteration cap, or —
// at the noise floor — a trial whose error equals the current one bit for bit): poses and points have not moved, so the
// squared errors of the previous pass 1 still stand for every measurement that survived the purge; only the histogram
// has to be rebuilt over them.  (The z <= 0 cases of that state were marked then and have been purged since.)
__global__ void __launch_bounds__(256) pass1_keep_kernel(BaDev d) {
    __shared__ unsigned hist[HIST_BINS];
    hist_clear(hist);
    constexpr int U = P1_U;
    for (int base = blockIdx.x * (256 * U); base < d.M; base += gridDim.x * (256 * U)) {
        int st[U];
        double e2[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int m = base + u * 256 + threadIdx.x, mc = min(m, d.M - 1);
            st[u] = m < d.M ? (int)d.m_state[mc] : (int)MS_DEAD;
            e2[u] = d.m_e2[mc];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (st[u] == MS_ALIVE) atomicAdd(&hist[e2_bin(e2[u])], 1u);
    }
    hist_flush(d, hist);
}

// The speculative step prologue's first launch: the purge that closes the finished step (guarded by end_step)
// and, if the loop goes on from an accepted trial (spec_go), pass 1 from the trial's errors — one dependent
// launch (~2.3 us) less than purge_kernel + pass1_from_trial_kernel.
__global__ void __launch_bounds__(256) purge_pass1_kernel(BaDev d) {
    TL_MARK(d, 1)
    if (d.sc->end_step == 0) return;
    const bool go = d.sc->spec_go != 0, stay = d.sc->spec_stay != 0;
    __shared__ unsigned hist[HIST_BINS];
    if (go || stay) hist_clear(hist);
    pass1_adopt<true>(d, go, go || stay, hist, stay);
    if (go || stay) hist_flush(d, hist);
}

// =================================================================================================
// K6: exact order statistic
// =================================================================================================
// histogram of an explicit key array (sharded mode: the gathered e^2 of all ranks)
__global__ void __launch_bounds__(256) hist_keys_kernel(const double* __restrict__ keys, long long n, unsigned* __restrict__ ghist) {
    __shared__ unsigned hist[HIST_BINS];
    for (int b = threadIdx.x; b < HIST_BINS; b += 256) hist[b] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        atomicAdd(&hist[e2_bin(keys[i])], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < HIST_BINS; b += 256) {
        const unsigned c = hist[b];
        if (c) atomicAdd(&ghist[b], c);
    }
}

// block-wide: locate the first-level bin that holds rank total/2 (every block computes the same
// answer from the global histogram; saves a launch + hand-off).  256 threads, 16 bins each.
__device__ void block_find_bin(const unsigned* __restrict__ hist, long long& total_out, int& bin_out, int& k_out) {
    __shared__ long long wsum[4];
    __shared__ long long s_total;
    __shared__ int s_bin, s_k;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    unsigned c[16];
    long long s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        c[j] = hist[16 * tid + j];
        s += c[j];
    }
    const long long incl = wave_incl_scan_i64(s);
    if (lane == 63) wsum[wid] = incl;
    if (tid == 0) {
        s_bin = -1;
        s_k = 0;
    }
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int i = 0; i < 4; i++) {
            const long long v = wsum[i];
            wsum[i] = run;
            run += v;
        }
        s_total = run;
    }
    __syncthreads();
    const long long total = s_total;
    const long long k = total / 2;   // vdErrorSquared[size()/2]
    const long long excl = wsum[wid] + incl - s;
    if (total > 0 && excl <= k && k < excl + s) {
        long long kk = k - excl;
        int bb = 16 * tid;
        for (int j = 0; j < 15; j++)
            if (kk >= c[j]) {
                kk -= c[j];
                bb++;
            } else
                break;
        s_bin = bb;
        s_k = (int)kk;
    }
    __syncthreads();
    total_out = total;
    bin_out = s_bin;
    k_out = s_k;
}

#define CAND_BUF 2048
// gather the keys of the selected first-level bin (keys = m_e2 masked by state, or an explicit
// array) and histogram their next 12 bits (47..36).  Candidates are staged in LDS and appended with
// one global atomic per flush.
__global__ void __launch_bounds__(256) select_compact_kernel(BaDev d, const double* __restrict__ keys, long long n,
                                                             const uint8_t* __restrict__ state) {
    TL_MARK(d, 2)
    if (ba_guard_blocks(d)) return;
    __shared__ double buf[CAND_BUF];
    __shared__ unsigned h2[HIST_BINS];
    __shared__ int cnt, gpos;
    const int tid = threadIdx.x;
    // the block's keys are fetched 4 per thread, unconditionally (clamped), and the first batch leaves BEFORE the
    // histogram scan below: one exposed round trip instead of state -> key per 256-key slice
    constexpr int U = 4;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long i0 = (long long)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    double key[U], nkey[U];
    bool ok[U], nok[U];
    auto fetch = [&](long long base, double* kq, bool* oq) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long i = base + u * 256 + tid;
            const long long ic = i < n ? i : (n > 0 ? n - 1 : 0);
            kq[u] = keys[ic];
            oq[u] = i < i1 && (!state || state[ic] == MS_ALIVE);
        }
    };
    fetch(i0, key, ok);
    long long total;
    int bin, k1;
    block_find_bin(d.hist, total, bin, k1);
    if (blockIdx.x == 0 && tid == 0) {
        d.sc->n_valid = total;
        d.sc->sel_bin = bin;
        d.sc->sel_k = k1;
    }
    if (bin < 0) return;
    for (int b = tid; b < HIST_BINS; b += 256) h2[b] = 0;
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (long long base = i0; base < i1; base += 256 * U) {
        const bool last = base + 256 * U >= i1;
        if (!last) fetch(base + 256 * U, nkey, nok);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (ok[u] && e2_bin(key[u]) == bin) {
                const int pos = atomicAdd(&cnt, 1);
                buf[pos] = key[u];
                atomicAdd(&h2[e2_bin2(key[u])], 1u);
            }
        __syncthreads();
        if (cnt > CAND_BUF - 256 * U || last) {
            const int c = cnt;
            if (tid == 0 && c > 0) gpos = atomicAdd(&d.sc->n_cand, c);
            __syncthreads();
            for (int j = tid; j < c; j += 256) d.cand[gpos + j] = buf[j];
            __syncthreads();
            if (tid == 0) cnt = 0;
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            key[u] = nkey[u];
            ok[u] = nok[u];
        }
    }
    unsigned* hist2 = d.hist + HIST_BINS;
    for (int b = tid; b < HIST_BINS; b += 256) {
        const unsigned c = h2[b];
        if (c) atomicAdd(&hist2[b], c);
    }
}

// MSB-first 8-bit radix select of rank k among src[0..n) restricted to keys whose bits above
// (first_shift + 8) equal those of `prefix`.  1024 threads.
__device__ unsigned long long block_radix_select(const double* src, int n, int k, unsigned long long prefix,
                                                 int first_shift, unsigned* hist, int* s_digit, int* s_k) {
    const int tid = threadIdx.x;
    for (int shift = first_shift; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(src[i]);
            // (shift + 8 == 64 on a full-width first pass: nothing is fixed yet, and a 64-bit shift by 64 is undefined)
            if (shift + 8 >= 64 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const int s = (int)(c0 + c1 + c2 + c3);
            const int incl = wave_incl_scan_i32(s);
            const int excl = incl - s;
            if (excl <= k && k < incl) {
                int kk = k - excl, dg = 4 * tid;
                if (kk >= (int)c0) {
                    kk -= c0;
                    dg++;
                    if (kk >= (int)c1) {
                        kk -= c1;
                        dg++;
                        if (kk >= (int)c2) {
                            kk -= c2;
                            dg++;
                        }
                    }
                }
                *s_digit = dg;
                *s_k = kk;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)(*s_digit) << shift;
        k = *s_k;
        __syncthreads();
    }
    return prefix;
}

#define SMALL_CAP 4096
// one block: second-level bin from hist2, collect its (few) members in LDS, finish the select there;
// sigma^2; reset both histograms
__global__ void __launch_bounds__(1024) select_final_kernel(BaDev d, int est, double min_sigma_sq) {
    TL_MARK(d, 3)
    if (ba_guard_blocks(d)) return;
    __shared__ double sm[SMALL_CAP];
    __shared__ unsigned hist[256];
    __shared__ long long wsum[16];
    __shared__ int s_digit, s_k, s_bin2, s_k2, s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = d.sc->n_cand;
    unsigned* hist2 = d.hist + HIST_BINS;
    unsigned long long result = 0;
    const int bin1 = d.sc->sel_bin;
    if (n > 0 && (bin1 == 0 || bin1 == HIST_BINS - 1)) {
        // clamped end bin: its members do not share their top bits; plain 8-pass select over them
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        result = block_radix_select(d.cand, n, d.sc->sel_k, 0ull, 56, hist, &s_digit, &s_k);
    } else if (n > 0) {
        // second-level bin
        unsigned c[4];
        long long s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            c[j] = hist2[4 * tid + j];
            s += c[j];
        }
        const long long incl = wave_incl_scan_i64(s);
        if (lane == 63) wsum[wid] = incl;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        if (tid == 0) {
            long long run = 0;
            for (int i = 0; i < 16; i++) {
                const long long v = wsum[i];
                wsum[i] = run;
                run += v;
            }
        }
        __syncthreads();
        const long long k1 = d.sc->sel_k;
        const long long excl = wsum[wid] + incl - s;
        if (excl <= k1 && k1 < excl + s) {
            long long kk = k1 - excl;
            int bb = 4 * tid;
            for (int j = 0; j < 3; j++)
                if (kk >= c[j]) {
                    kk -= c[j];
                    bb++;
                } else
                    break;
            s_bin2 = bb;
            s_k2 = (int)kk;
        }
        __syncthreads();
        const int bin2 = s_bin2, k2 = s_k2;
        for (int i = tid; i < n; i += 1024) {
            const double key = d.cand[i];
            if (e2_bin2(key) == bin2) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < SMALL_CAP) sm[pos] = key;
            }
        }
        __syncthreads();
        const int m = s_cnt;
        // bits 63..36 are fixed by (bin, bin2)
        const unsigned long long top = ((unsigned long long)(bin1 + E2_BIN_BASE) << 48) | ((unsigned long long)bin2 << 36);
        if (m <= 128) {
            // the usual case, a handful of keys share both bins: rank by counting, one barrier (five radix passes with
            // three barriers each are ~1 us of this single-workgroup kernel).  Keys are positive doubles: their bit
            // patterns order like the values; ties are broken by position, so exactly one key has rank k2.
            __shared__ unsigned long long s_res;
            if (tid < m) {
                const unsigned long long mine = (unsigned long long)__double_as_longlong(sm[tid]);
                int rank = 0;
                for (int j = 0; j < m; j++) {
                    const unsigned long long o = (unsigned long long)__double_as_longlong(sm[j]);
                    rank += (o < mine || (o == mine && j < tid)) ? 1 : 0;
                }
                if (rank == k2) s_res = mine;
            }
            __syncthreads();
            result = s_res;
        } else if (m <= SMALL_CAP)
            result = block_radix_select(sm, m, k2, top, 32, hist, &s_digit, &s_k);
        else   // pathological (thousands of near-identical keys): same passes straight from global
            result = block_radix_select(d.cand, n, k2, top, 32, hist, &s_digit, &s_k);
    }
    for (int b = tid; b < 2 * HIST_BINS; b += 1024) d.hist[b] = 0;
    if (tid == 0) {
        const double med = n > 0 ? __longlong_as_double((long long)result) : 0.0;
        d.sc->median = med;
        double s2 = est_sigma_sq_from_median(est, med, (unsigned long long)d.sc->n_valid);
        if (s2 < min_sigma_sq) s2 = min_sigma_sq;   // :234-237
        d.sc->sigma_sq = s2;
        d.sc->n_bad = 0;
        d.sc->n_cand = 0;
    }
}

// ---- sharded exact select: three small all-reduces, no host round trip ---------------------------------
// Every rank histograms its own e^2; the first-level histogram is all-reduced, every rank finds the same
// bin and compacts its own candidates + second-level histogram; that is all-reduced too; the few keys
// that share both bins (a 2^-16 relative window around the order statistic) are exchanged through
// fixed-size per-rank slots of a zero-initialised buffer (sum all-reduce = concatenation), and every rank
// finishes with the same radix select.  Histograms travel as doubles (the hook reduces fp64; counts are
// exact).  A rank with more last-stage candidates than its slot holds raises select_overflow: the host
// then repeats the LM step with the gather-everything path (thousands of bit-identical errors only).
#define XCAND_CAP_DEFAULT 1024   // keys per rank slot (a 2^-16 relative window around the median holds ~1e-5 of the keys) (PTAM_XCAND_CAP overrides it: the tests force the overflow path)
__global__ void hist_to_f64_kernel(const unsigned* __restrict__ h, double* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)h[i];
}
__global__ void f64_to_hist_kernel(const double* __restrict__ in, unsigned* __restrict__ h, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) h[i] = (unsigned)(in[i] + 0.5);
}

// stage 2: second-level bin from the all-reduced histogram; my candidates of that bin go into my slot.
// xchg: [world counts][world x cap keys], zeroed beforehand.
__global__ void __launch_bounds__(1024) select_stage_kernel(BaDev d, double* __restrict__ xchg, int rank, int world, int XCAND_CAP) {
    __shared__ long long wsum[16];
    __shared__ int s_bin2, s_k2, s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = d.sc->n_cand;
    const int bin1 = d.sc->sel_bin;
    const unsigned* hist2 = d.hist + HIST_BINS;
    const bool clamped = bin1 == 0 || bin1 == HIST_BINS - 1;   // members do not share their top bits: all of them count
    if (tid == 0) {
        s_cnt = 0;
        s_bin2 = -1;
        s_k2 = d.sc->sel_k;
    }
    unsigned c[4];
    long long s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c[j] = hist2[4 * tid + j];
        s += c[j];
    }
    const long long incl = wave_incl_scan_i64(s);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int i = 0; i < 16; i++) {
            const long long v = wsum[i];
            wsum[i] = run;
            run += v;
        }
    }
    __syncthreads();
    if (bin1 >= 0 && !clamped) {
        const long long k1 = d.sc->sel_k;
        const long long excl = wsum[wid] + incl - s;
        if (excl <= k1 && k1 < excl + s) {
            long long kk = k1 - excl;
            int bb = 4 * tid;
            for (int j = 0; j < 3; j++)
                if (kk >= c[j]) {
                    kk -= c[j];
                    bb++;
                } else
                    break;
            s_bin2 = bb;
            s_k2 = (int)kk;
        }
    }
    __syncthreads();
    const int bin2 = s_bin2;
    double* slot = xchg + world + (size_t)rank * XCAND_CAP;
    if (bin1 >= 0)
        for (int i = tid; i < n; i += 1024) {
            const double key = d.cand[i];
            if (clamped || e2_bin2(key) == bin2) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < XCAND_CAP) slot[pos] = key;
            }
        }
    __syncthreads();
    if (tid == 0) {
        xchg[rank] = (double)s_cnt;
        d.sc->sel_bin2 = bin2;
        d.sc->sel_k2 = s_k2;
    }
}

// stage 3: concatenate the exchanged candidates, select, derive sigma^2 (tail of select_final_kernel)
__global__ void __launch_bounds__(1024) select_finish_kernel(BaDev d, const double* __restrict__ xchg, double* __restrict__ list,
                                                             int world, int XCAND_CAP, int est, double min_sigma_sq) {
    __shared__ unsigned hist[256];
    __shared__ int s_digit, s_k, s_over;
    __shared__ int offs[34];
    const int tid = threadIdx.x;
    if (tid == 0) {
        int run = 0, over = 0;
        for (int r = 0; r < world; r++) {
            int c = (int)(xchg[r] + 0.5);
            if (c > XCAND_CAP) {
                over = 1;
                c = XCAND_CAP;
            }
            offs[r] = run;
            run += c;
        }
        offs[world] = run;
        s_over = over;
    }
    __syncthreads();
    const int total = offs[world];
    for (int r = 0; r < world; r++) {
        const int c = offs[r + 1] - offs[r];
        const double* slot = xchg + world + (size_t)r * XCAND_CAP;
        for (int i = tid; i < c; i += 1024) list[offs[r] + i] = slot[i];
    }
    __syncthreads();
    const int bin1 = d.sc->sel_bin, bin2 = d.sc->sel_bin2;
    unsigned long long result = 0;
    const bool any = bin1 >= 0 && total > 0;
    if (any) {
        if (bin2 < 0)
            result = block_radix_select(list, total, d.sc->sel_k2, 0ull, 56, hist, &s_digit, &s_k);
        else {
            const unsigned long long top = ((unsigned long long)(bin1 + E2_BIN_BASE) << 48) | ((unsigned long long)bin2 << 36);
            result = block_radix_select(list, total, d.sc->sel_k2, top, 32, hist, &s_digit, &s_k);
        }
    }
    for (int b = tid; b < 2 * HIST_BINS; b += 1024) d.hist[b] = 0;
    if (tid == 0) {
        const double med = any ? __longlong_as_double((long long)result) : 0.0;
        d.sc->median = med;
        double s2 = est_sigma_sq_from_median(est, med, (unsigned long long)d.sc->n_valid);
        if (s2 < min_sigma_sq) s2 = min_sigma_sq;   // :234-237
        d.sc->sigma_sq = s2;
        d.sc->n_bad = 0;
        d.sc->n_cand = 0;
        d.sc->select_overflow = s_over;
    }
}

// compact this rank's valid e^2 (sharded mode)
__global__ void __launch_bounds__(256) compact_valid_kernel(BaDev d, double* __restrict__ out, int* __restrict__ counter) {
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * 256; base < d.M; base += gridDim.x * 256) {
        const int i = base + threadIdx.x;
        const bool take = i < d.M && d.m_state[i] == MS_ALIVE;
        const unsigned long long m = __ballot(take);
        if (m) {
            int pos = 0;
            if (lane == 0) pos = atomicAdd(counter, __popcll(m));
            pos = __shfl(pos, 0, 64);
            if (take) out[pos + __popcll(m & ((1ull << lane) - 1ull))] = d.m_e2[i];
        }
    }
}

// =================================================================================================
// K7: fused Jacobian + normal-equation accumulation (pass 2, :250-332)
// =================================================================================================
// One measurement of pass 2: weight, Jacobians, camera accumulation (LDS atomics into Ul), W store.
// Returns B (2x3) and the weighted error for the caller's per-point reduction.
__device__ __forceinline__ void jac_measure(const DevCam& cam, const BaDev& d, const double* __restrict__ pose,
                                            const double* __restrict__ pt, double sigma_sq, double inv_sigma_sq, int est,
                                            int m, double* Ul,
                                            double& err, int& nbad, double B0[3], double B1[3], double& ex, double& ey) {
    double Wv[18];
#pragma unroll
    for (int k = 0; k < 18; k++) Wv[k] = 0;
    const int st = d.m_state[m];
    if (st == MS_BAD) {   // z <= 0 in pass 1  (:259-263)
        err += 1.0;
        nbad++;
    } else if (st == MS_ALIVE) {
        const int c = d.m_cam[m], p = d.m_pt[m];
        const double* __restrict__ T = pose + 12 * c;
        BaProj pr;
        ba_project(cam, T, pt + 3 * p, pr);
        const double X = pr.X, Y = pr.Y, Z = pr.Z;
        const double2 fo = d.m_found[m];
        const double s = d.m_s[m];
        ex = s * (fo.x - pr.u);
        ey = s * (fo.y - pr.v);
        const double e2 = ex * ex + ey * ey;
        const double w = ba_sqrt_weight(est, e2, sigma_sq, inv_sigma_sq);
        ex *= w;   // meas.v2Epsilon = dWeight * meas.v2Epsilon  (:272)
        ey *= w;
        if (w == 0) {   // :274-279
            d.m_state[m] = MS_BAD;
            err += 1.0;
            nbad++;
            ex = ey = 0;
        } else {
            err += ba_objective(est, e2, sigma_sq, inv_sigma_sq);
            double D[4];
            ba_derivs(cam, pr, D);
            // fold sqrt-weight and dSqrtInvNoise into the camera derivatives (:285, :302)
            const double D0 = s * w * D[0], D1 = s * w * D[1], D2 = s * w * D[2], D3 = s * w * D[3];
            const double iz = rcp_nr(Z);
            // B: point Jacobian, motion = m-th column of R_cw (:306-313)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double g0 = T[k], g1 = T[3 + k], g2 = T[6 + k];
                const double mx = (g0 - X * g2 * iz) * iz, my = (g1 - Y * g2 * iz) * iz;
                B0[k] = D0 * mx + D1 * my;
                B1[k] = D2 * mx + D3 * my;
            }
            const int fidx = d.m_fidx[m];
            if (fidx >= 0) {
                // A: camera Jacobian, SE3 generator fields (:291-303)
                const double gx[6] = {1, 0, 0, 0, Z, -Y};
                const double gy[6] = {0, 1, 0, -Z, 0, X};
                const double gz[6] = {0, 0, 1, Y, -X, 0};
                double A0[6], A1[6];
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const double mx = (gx[k] - X * gz[k] * iz) * iz, my = (gy[k] - Y * gz[k] * iz) * iz;
                    A0[k] = D0 * mx + D1 * my;
                    A1[k] = D2 * mx + D3 * my;
                }
                double* Uc = Ul + fidx * 27;
                int k = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = 0; b <= a; b++) atomicAdd(&Uc[k++], A0[a] * A0[b] + A1[a] * A1[b]);   // U_LL :21-26
#pragma unroll
                for (int a = 0; a < 6; a++) atomicAdd(&Uc[21 + a], A0[a] * ex + A1[a] * ey);       // epsA :321
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) Wv[a * 3 + b] = A0[a] * B0[b] + A1[a] * B1[b];   // W = A^T B :331
            }
        }
    }
    // W planes: 9 x double2, coalesced across the wave
#pragma unroll
    for (int k = 0; k < 9; k++) d.W[(size_t)k * d.M + m] = make_double2(Wv[2 * k], Wv[2 * k + 1]);
}

// flush a workgroup's camera partials + error partial (end of both accumulate kernels)
template <int THREADS>
__device__ __forceinline__ void jac_flush(const BaDev& d, const double* Ul, double err, int nbad) {
    __shared__ double werr[THREADS / 64];
    __shared__ int wbad[THREADS / 64];
    const int tid = threadIdx.x;
    double* up = d.Upart + (size_t)blockIdx.x * d.F * 27;
    for (int k = tid; k < d.F * 27; k += THREADS) up[k] = Ul[k];
    err = wave_sum_f64(err);
    nbad = wave_sum_i32(nbad);
    if ((tid & 63) == 0) {
        werr[tid >> 6] = err;
        wbad[tid >> 6] = nbad;
    }
    __syncthreads();
    if (tid == 0) {
        double e = 0;
        int b = 0;
        for (int i = 0; i < THREADS / 64; i++) {
            e += werr[i];
            b += wbad[i];
        }
        d.err_part[2 * blockIdx.x] = e;
        d.bad_part[blockIdx.x] = b;
    }
}

// K7, wave variant (every point has <= 64 measurements): lane = measurement, a wave takes runs of
// 64 consecutive measurements of the point-major order.  No workgroup barrier inside the loop.
//  - balance: every wave gets `per_wave` (or one more) consecutive 64-measurement chunks, and
//    the next chunk's inputs are prefetched while the current one is computed;
//  - poses are staged once per workgroup in LDS (C*96 B); the chunk's points are fetched one per
//    lane and handed to their measurements by ds_bpermute: ONE global round trip per chunk;
//  - U / epsA: ds_add_f64 into the workgroup's LDS partials;
//  - V / epsB: segmented inclusive scan over the wave (DPP row shifts + row broadcasts, fixed tree).
//    A point lying inside the chunk is stored by its last lane; a point cut by a chunk boundary leaves
//    one piece per chunk in d.cut (plain stores) and vinv_kernel adds them in chunk order — no global
//    atomics (device-scope fp64 atomics cost this launch ~1.9 us at 50 x 5000), no zeroing pass;
//  - W: 9 coalesced double2 planes.
// dynamic LDS: Ul[F*27] | poses[C*12] + 1 spare slot
struct K7In {
    int st, c, p, fidx;
    double2 fo;
    double sn;
    double px, py, pz;   // point (pt0 + lane), fetched one per lane
    int pt0;
    int p_prev, p_next;  // point of the measurement just before / after this chunk (-1 at the ends)
};
__device__ __forceinline__ void k7_load(const BaDev& d, const double* __restrict__ pt, int m0, int lane, K7In& in) {
    // every load is unconditional (clamped index, result masked afterwards): with no branch between
    // them the compiler can wait for the older ones by count instead of draining the whole queue
    const int m = m0 + lane;
    const int mlast = min(m0 + 63, d.M - 1);
    const int mc = min(m, mlast);
    in.pt0 = d.m_pt[m0];
    const int pt1 = d.m_pt[mlast];
    in.p_prev = d.m_pt[max(m0 - 1, 0)];
    in.p_next = d.m_pt[min(m0 + 64, d.M - 1)];
    if (m0 == 0) in.p_prev = -1;
    if (m0 + 64 >= d.M) in.p_next = -1;
    in.st = d.m_state[mc];
    in.c = d.m_cam[mc];
    in.p = d.m_pt[mc];
    in.fidx = d.m_fidx[mc];
    in.fo = d.m_found[mc];
    in.sn = d.m_s[mc];
    // (lanes past the end hold a copy of the last measurement: the caller masks them)
    const double* q = pt + 3 * (size_t)min(in.pt0 + lane, pt1);
    in.px = q[0];
    in.py = q[1];
    in.pz = q[2];
}

#ifdef K7_TIMING
#define K7_STAMP(i) if (blockIdx.x == 7 && tid == 0) d.dbg[i] = (long long)__builtin_amdgcn_s_memrealtime();
#define K7_WALL(i) if (tid == 0 && blockIdx.x < 2000) d.dbg[16 + 2 * blockIdx.x + i] = (long long)__builtin_amdgcn_s_memrealtime();
#else
#define K7_WALL(i)
#define K7_STAMP(i)
#endif
// EST: the M-estimator as a compile-time constant (-1: taken from `est_arg`) — the default Tukey path then
// carries none of the Cauchy / Huber code (log, sqrt and their constants)
// ablation switches for tools/ experiments (never defined in the product build)
#ifdef K7_NOATOM
#define K7_UADD(p, v) asm volatile("" ::"v"(v))
#else
#define K7_UADD(p, v) atomicAdd(p, v)
#endif
#ifdef K7_NOW
#define K7_WSTORE(dst, v) { const double2 v_ = v; asm volatile("" ::"v"(v_.x), "v"(v_.y)); }
#else
// W is written once and read by the NEXT kernel from every XCD: write-through (sc1) 16-byte stores leave no dirty
// lines for the end-of-kernel L2 write-back (measured -7 % launch time at 50 x 5000; `nt` is slower).
// The s_nop covers the gfx940+ hazard the compiler cannot see inside the asm: a VALU write of the data registers
// of a > 64-bit global store needs two wait states after it.
typedef double k7_d2 __attribute__((ext_vector_type(2)));
#define K7_WSTORE(dst, v) { const double2 v_ = v; const k7_d2 w_ = {v_.x, v_.y}; asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(&(dst)), "v"(w_) : "memory"); }
#endif
template <int THREADS, bool PREFETCH, bool LOOP, int EST>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 8)))
jac_accum_wave_kernel(DevCam cam, BaDev d, int cur, int est_arg, int per_wave, int extra) {
    TL_MARK(d, 4)
    if (ba_guard_blocks(d)) return;
    if (d.guard == 1 && d.sc->spec_stay) cur ^= 1;   // (launched for the trial state; "stay": the unchanged current one)
    const int est = EST >= 0 ? EST : est_arg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ul = smem;
    double* Ps = smem + (((size_t)d.F * 27 + 1) & ~(size_t)1);
    const int tid = threadIdx.x, lane = tid & 63;
    K7_WALL(0)
    const double* __restrict__ pt = d.pt[cur];
    const int n_chunks64 = (d.M + 63) >> 6;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: chunk bases live in SGPRs
    // chunk range of this wave: `per_wave` chunks each, and the first `extra` waves in (wave-in-block major,
    // block minor) order take one more — the surplus lands on different CUs / SIMDs instead of piling
    // onto the leading workgroups.  Ranges stay contiguous and ordered by (block, wave).
    int c_begin, c_end;
    {
        constexpr int WPB = THREADS / 64;
        const int blk = blockIdx.x, grid = gridDim.x;
        const int full = extra / grid, rem = extra - full * grid;   // waves with wid < full (or == full, blk < rem) are long
        const int before = blk * full + min(blk, rem) + min(wid, full) + ((wid > full && blk < rem) ? 1 : 0);
        const int mine = per_wave + ((wid < full || (wid == full && blk < rem)) ? 1 : 0);
        c_begin = (blk * WPB + wid) * per_wave + before;
        c_end = min(n_chunks64, c_begin + mine);
    }
    // Global loads leave in the order they are needed: poses (staged to LDS before the barrier), then
    // the first chunk's inputs and sigma^2 — memory returns in order, so the barrier waits for ONE
    // round trip while the chunk's second, dependent one (m_pt -> point) is still in flight.
    for (int k = tid; k < d.F * 27; k += THREADS) Ul[k] = 0;
    const double* __restrict__ pose = d.pose[cur];
    const int n_pose = d.C * 12;
    double pv[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int k = tid + i * THREADS;
        pv[i] = pose[min(k, n_pose - 1)];
    }
    K7In in;
    if (PREFETCH) k7_load(d, pt, min(c_begin, n_chunks64 - 1) << 6, lane, in);   // (idle waves load a valid chunk and drop it)
    const double sigma_sq = d.sc->sigma_sq;
#pragma unroll
    for (int i = 0; i < 2; i++) {   // (unconditional like the loads: surplus threads hit the spare slot Ps[n_pose])
        const int k = tid + i * THREADS;
        Ps[min(k, n_pose)] = pv[i];
    }
    for (int k = tid + 2 * THREADS; k < n_pose; k += THREADS) Ps[k] = pose[k];
    __syncthreads();
    K7_STAMP(0)
    const double inv_sigma_sq = 1.0 / sigma_sq;
    double err = 0;
    int nbad = 0;
    K7_STAMP(1)
#pragma unroll 1
    for (int ci = c_begin; ci < c_end; ci++) {   // (LOOP == false: exactly one chunk per wave, see the break below)
        const int m0 = ci << 6;
        const int m = m0 + lane;
        const bool active = m < d.M;
        if (!PREFETCH) k7_load(d, pt, m0, lane, in);
        const K7In cu = in;
        K7_STAMP(2)
        if (PREFETCH && LOOP && ci + 1 < c_end) k7_load(d, pt, (ci + 1) << 6, lane, in);   // prefetch the next chunk
        const int st = active ? cu.st : MS_DEAD, c = cu.c, p = cu.p, fidx = active ? cu.fidx : -1;
        const int src = p - cu.pt0;   // lane that fetched my point
        const double Xw = __shfl(cu.px, src, 64), Yw = __shfl(cu.py, src, 64), Zw = __shfl(cu.pz, src, 64);
        int pid = active ? p : (-1 - lane);   // inactive lanes: unique ids, never merged
        double v[9];
#pragma unroll
        for (int i = 0; i < 9; i++) v[i] = 0;
        bool full = false;
        BaProj pr;
        double ex = 0, ey = 0, w = 0;
        const double* T = Ps + 12 * c;
        if (st == MS_BAD) {   // z <= 0 in pass 1  (:259-263)
            err += 1.0;
            nbad++;
        } else if (st == MS_ALIVE) {
            const double Xp[3] = {Xw, Yw, Zw};
            ba_project(cam, T, Xp, pr);
            ex = cu.sn * (cu.fo.x - pr.u);
            ey = cu.sn * (cu.fo.y - pr.v);
            const double e2 = ex * ex + ey * ey;
            w = ba_sqrt_weight(est, e2, sigma_sq, inv_sigma_sq);
            if (w == 0) {   // :274-279
                d.m_state[m] = MS_BAD;
                err += 1.0;
                nbad++;
            } else {
                err += ba_objective(est, e2, sigma_sq, inv_sigma_sq);
                full = true;
            }
        }
        K7_STAMP(3)
        if (full) {
            ex *= w;   // meas.v2Epsilon = dWeight * meas.v2Epsilon  (:272)
            ey *= w;
            double D[4];
            ba_derivs(cam, pr, D);
            const double sw = cu.sn * w;   // fold sqrt-weight and dSqrtInvNoise into the derivatives (:285, :302)
            const double D0 = sw * D[0], D1 = sw * D[1], D2 = sw * D[2], D3 = sw * D[3];
            const double X = pr.X, Y = pr.Y, Z = pr.Z;
            const double iz = rcp_nr(Z);
            __builtin_amdgcn_sched_barrier(0);   // phase fences keep live ranges short (128-VGPR budget)
            double B0[3], B1[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {   // B: motion = k-th column of R_cw (:306-313)
                const double g0 = T[k], g1 = T[3 + k], g2 = T[6 + k];
                const double mx = (g0 - X * g2 * iz) * iz, my = (g1 - Y * g2 * iz) * iz;
                B0[k] = D0 * mx + D1 * my;
                B1[k] = D2 * mx + D3 * my;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (fidx >= 0) {
                // A: SE3 generator fields (:291-303): e_k for k<3 ; (0,-Z,Y) (Z,0,-X) (-Y,X,0)
                const double zx = -X * iz * iz, zy = -Y * iz * iz;   // d(x)/dZ, d(y)/dZ
                double A0[6], A1[6];
                {
                    const double mxs[6] = {iz, 0, zx, zx * Y, iz * Z - zx * X, -iz * Y};
                    const double mys[6] = {0, iz, zy, -iz * Z + zy * Y, -zy * X, iz * X};
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        A0[k] = D0 * mxs[k] + D1 * mys[k];
                        A1[k] = D2 * mxs[k] + D3 * mys[k];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                K7_STAMP(4)
                double* Uc = Ul + fidx * 27;
                int k = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = 0; b <= a; b++) K7_UADD(&Uc[k++], A0[a] * A0[b] + A1[a] * A1[b]);   // U_LL :21-26
#pragma unroll
                for (int a = 0; a < 6; a++) K7_UADD(&Uc[21 + a], A0[a] * ex + A1[a] * ey);       // epsA :321
                __builtin_amdgcn_sched_barrier(0);
                K7_STAMP(5)
#pragma unroll
                for (int q = 0; q < 9; q++) {   // W = A^T B (:331), 9 coalesced double2 planes
                    const int i0 = 2 * q, i1 = 2 * q + 1;
                    K7_WSTORE(d.W[(size_t)q * d.M + m], make_double2(A0[i0 / 3] * B0[i0 % 3] + A1[i0 / 3] * B1[i0 % 3],
                                                                     A0[i1 / 3] * B0[i1 % 3] + A1[i1 / 3] * B1[i1 % 3]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            v[0] = B0[0] * B0[0] + B1[0] * B1[0];   // V_LL 00,10,11,20,21,22  (:325)
            v[1] = B0[1] * B0[0] + B1[1] * B1[0];
            v[2] = B0[1] * B0[1] + B1[1] * B1[1];
            v[3] = B0[2] * B0[0] + B1[2] * B1[0];
            v[4] = B0[2] * B0[1] + B1[2] * B1[1];
            v[5] = B0[2] * B0[2] + B1[2] * B1[2];
            v[6] = B0[0] * ex + B1[0] * ey;         // epsB (:326)
            v[7] = B0[1] * ex + B1[1] * ey;
            v[8] = B0[2] * ex + B1[2] * ey;
        }
        if (active && !(full && fidx >= 0)) {
            const double2 z2 = make_double2(0, 0);
#pragma unroll
            for (int q = 0; q < 9; q++) d.W[(size_t)q * d.M + m] = z2;
        }
        // segmented inclusive scan keyed by the point id: 4 in-row steps (DPP row_shr 1,2,4,8), then
        // the row carries (row_bcast15 into rows 1,3; row_bcast31 into rows 2,3).  Segments are
        // contiguous, so "the source lane has my point id" implies every lane in between has it too.
        K7_STAMP(6)
        // ids are compared as pid + 1 so that the 0 a row-shift returns for "no source lane" never
        // matches a real point (inactive lanes carry zeros, a spurious match there adds 0)
        const int pid1 = pid + 1;
#define SEG_STEP(GETP, GETV)                                             \
    {                                                                    \
        const int po = GETP;                                             \
        double t[9];                                                     \
        _Pragma("unroll") for (int i = 0; i < 9; i++) t[i] = GETV;       \
        if (po == pid1) {                                                \
            _Pragma("unroll") for (int i = 0; i < 9; i++) v[i] += t[i];  \
        }                                                                \
    }
#ifndef K7_NOSCAN
        SEG_STEP(dpp_row_shr0_i32<1>(pid1), dpp_row_shr_f64<1>(v[i]))
        SEG_STEP(dpp_row_shr0_i32<2>(pid1), dpp_row_shr_f64<2>(v[i]))
        SEG_STEP(dpp_row_shr0_i32<4>(pid1), dpp_row_shr_f64<4>(v[i]))
        SEG_STEP(dpp_row_shr0_i32<8>(pid1), dpp_row_shr_f64<8>(v[i]))
        // row carries: only rows 1,3 (then 2,3) take part; 0 never matches a point id
        SEG_STEP((dpp_bcastx_i32<0x142>(pid1) & -((lane >> 4) & 1)), dpp_bcastx_f64<0x142>(v[i]))
        SEG_STEP((dpp_bcastx_i32<0x143>(pid1) & -((lane >> 5) & 1)), dpp_bcastx_f64<0x143>(v[i]))
#endif
#undef SEG_STEP
        K7_STAMP(7)
        const int pn = __shfl_down(pid, 1, 64);
        if (active && (lane == 63 || pn != pid)) {
            // a point cut by a chunk boundary leaves its piece in the chunk's slot (leading segment: 2c, trailing:
            // 2c + 1); vinv_kernel adds the pieces in chunk order — plain stores, no zeroing pass, fixed order
            const bool whole = pid != cu.p_prev && pid != cu.p_next;
            double* Vp = whole ? d.V + (size_t)pid * 6 : d.cut + (size_t)(2 * ci + (pid == cu.p_prev ? 0 : 1)) * 9;
            double* Ep = whole ? d.epsB + (size_t)pid * 3 : Vp + 6;
#pragma unroll
            for (int i = 0; i < 6; i++) Vp[i] = v[i];
            Ep[0] = v[6];
            Ep[1] = v[7];
            Ep[2] = v[8];
        }
        if (!LOOP) break;
    }
    K7_STAMP(8)
    __syncthreads();
    jac_flush<THREADS>(d, Ul, err, nbad);
    K7_STAMP(9)
    K7_WALL(1)
}

// the wave-variant instantiations: {one chunk per wave | looping, 256 or 512 threads} x {Tukey | run-time estimator}
static const void* k7_wave_fn(int threads, bool loop, int est) {
    const bool tukey = est == PTAM_EST_TUKEY;
    if (!loop)
        return tukey ? (const void*)jac_accum_wave_kernel<512, true, false, PTAM_EST_TUKEY>
                     : (const void*)jac_accum_wave_kernel<512, true, false, -1>;
    if (threads == 256)
        return tukey ? (const void*)jac_accum_wave_kernel<256, true, true, PTAM_EST_TUKEY>
                     : (const void*)jac_accum_wave_kernel<256, false, true, -1>;
    return tukey ? (const void*)jac_accum_wave_kernel<512, true, true, PTAM_EST_TUKEY>
                 : (const void*)jac_accum_wave_kernel<512, false, true, -1>;
}

// K7, block variant (points with up to BA_CHUNK measurements): a workgroup owns whole points.
// dynamic LDS: Ul[F*27] camera partials | Bs[BA_CHUNK][8] per-measurement B (2x3) and weighted eps
__global__ void __launch_bounds__(BA_CHUNK) jac_accum_kernel(DevCam cam, BaDev d, int cur, int est) {
    if (ba_guard_blocks(d)) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ul = smem;
    double* Bs = smem + (((size_t)d.F * 27 + 1) & ~(size_t)1);
    const int tid = threadIdx.x;
    for (int k = tid; k < d.F * 27; k += BA_CHUNK) Ul[k] = 0;
    __syncthreads();
    const double* __restrict__ pose = d.pose[cur];
    const double* __restrict__ pt = d.pt[cur];
    const double sigma_sq = d.sc->sigma_sq;
    const double inv_sigma_sq = 1.0 / sigma_sq;
    double err = 0;
    int nbad = 0;
    for (int ci = blockIdx.x; ci < d.n_chunks; ci += gridDim.x) {
        const BaChunk ch = d.chunks[ci];
        const int m = ch.m_begin + tid;
        double B0[3] = {0, 0, 0}, B1[3] = {0, 0, 0}, ex = 0, ey = 0;
        if (m < ch.m_end) jac_measure(cam, d, pose, pt, sigma_sq, inv_sigma_sq, est, m, Ul, err, nbad, B0, B1, ex, ey);
        double* bs = Bs + tid * 8;
        bs[0] = B0[0];
        bs[1] = B0[1];
        bs[2] = B0[2];
        bs[3] = B1[0];
        bs[4] = B1[1];
        bs[5] = B1[2];
        bs[6] = ex;
        bs[7] = ey;
        __syncthreads();
        // V_LL += B^T B, epsB += B^T eps : this block owns every measurement of its points, so the
        // sums are complete and written once, in measurement order (deterministic)  (:325-326)
        const int npts = ch.pt_end - ch.pt_begin;
        for (int task = tid; task < npts * 9; task += BA_CHUNK) {
            const int pi = task / 9, o = task - pi * 9;
            const int p = ch.pt_begin + pi;
            const int r0 = d.rowptr[p] - ch.m_begin, r1 = d.rowptr[p + 1] - ch.m_begin;
            double acc = 0;
            if (o < 6) {
                const int a = o < 1 ? 0 : (o < 3 ? 1 : 2);
                const int b = o - (a * (a + 1)) / 2;
                for (int rr = r0; rr < r1; rr++) {
                    const double* q = Bs + rr * 8;
                    acc += q[a] * q[b] + q[3 + a] * q[3 + b];
                }
                d.V[(size_t)p * 6 + o] = acc;
            } else {
                const int a = o - 6;
                for (int rr = r0; rr < r1; rr++) {
                    const double* q = Bs + rr * 8;
                    acc += q[a] * q[6] + q[3 + a] * q[7];
                }
                d.epsB[(size_t)p * 3 + a] = acc;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    jac_flush<BA_CHUNK>(d, Ul, err, nbad);
}

// fixed-order sum over the accumulate grid.  Stage A (this kernel): grid (column groups of 64,
// RSPLIT row splits); wave w of a block sums rows w, w+4, ... of its split with coalesced 512-byte
// loads, the four waves combine in fixed order -> Usplit[split][F*27].  Stage B is folded into the
// consumers (schur_reduce_kernel sums the RSPLIT values of the 27 numbers it needs per camera).
// Block (0,0) also reduces the error / bad-count partials.
#define RSPLIT 16
__device__ __forceinline__ void reduce_partials_body(const BaDev& d, int grid_acc, int bx, int by) {
    __shared__ double comb[4][64];
    __shared__ double werr[4];
    __shared__ int wbad[4];
    const int total = d.F * 27;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = bx * 64 + lane;
    const int split = by;
    const int rows_per = (grid_acc + RSPLIT - 1) / RSPLIT;
    const int r0 = split * rows_per, r1 = min(grid_acc, r0 + rows_per);
    double s = 0;
    if (col < total)
        for (int b = r0 + wid; b < r1; b += 4) s += d.Upart[(size_t)b * total + col];
    comb[wid][lane] = s;
    __syncthreads();
    if (wid == 0 && col < total) d.Usplit[(size_t)split * total + col] = ((comb[0][lane] + comb[1][lane]) + comb[2][lane]) + comb[3][lane];
    if (bx == 0 && by == 0) {
        double e = 0;
        int nb = 0;
        for (int b = threadIdx.x; b < grid_acc; b += 256) {
            e += d.err_part[2 * b];
            nb += d.bad_part[b];
        }
        e = wave_sum_f64(e);
        nb = wave_sum_i32(nb);
        if (lane == 0) {
            werr[wid] = e;
            wbad[wid] = nb;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            d.sc->cur_err = ((werr[0] + werr[1]) + werr[2]) + werr[3];
            d.sc->n_bad = wbad[0] + wbad[1] + wbad[2] + wbad[3];
        }
    }
}

__global__ void __launch_bounds__(256) reduce_partials_kernel(BaDev d, int grid_acc) {
    TL_MARK(d, 16)
    if (ba_guard_blocks(d)) return;
    reduce_partials_body(d, grid_acc, blockIdx.x, blockIdx.y);
}

// =================================================================================================
// K8a: V*^-1  (:341-359)   TooN Cholesky<3>::get_inverse
// =================================================================================================
__device__ __forceinline__ void vinv_body(const BaDev& d, double lambda, int bx) {
    const int p = bx * 256 + threadIdx.x;
    if (p >= d.P) return;
    double v[6];
    const int r0 = d.rowptr[p], r1 = d.rowptr[p + 1];
    const int c0 = r0 >> 6, c1 = (r1 - 1) >> 6;
    if (d.cut != nullptr && r1 > r0 && c0 != c1) {
        // the point's measurements span chunks c0..c1 of K7's wave variant: trailing segment of c0, then the leading
        // segments of c0+1..c1, added in that order; epsB is completed here for the Schur / back-substitution kernels
        const double* q = d.cut + (size_t)(2 * c0 + 1) * 9;
        double e[3] = {q[6], q[7], q[8]};
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] = q[i];
        for (int c = c0 + 1; c <= c1; c++) {
            q = d.cut + (size_t)(2 * c) * 9;
#pragma unroll
            for (int i = 0; i < 6; i++) v[i] += q[i];
#pragma unroll
            for (int i = 0; i < 3; i++) e[i] += q[6 + i];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) d.epsB[(size_t)p * 3 + i] = e[i];
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] = d.V[(size_t)p * 6 + i];
    }
    double A[9] = {v[0], v[1], v[3], v[1], v[2], v[4], v[3], v[4], v[5]};
    double* out = d.Vinv + (size_t)p * 9;
    if (A[0] * A[4] * A[8] == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) out[i] = 0;
        return;
    }
    A[0] *= (1.0 + lambda);
    A[4] *= (1.0 + lambda);
    A[8] *= (1.0 + lambda);
    // LDL^T, lower triangle; strict upper caches the undivided column
    for (int col = 0; col < 3; col++) {
        double inv_diag = 1;
        for (int row = col; row < 3; row++) {
            double val = A[row * 3 + col];
            for (int c2 = 0; c2 < col; c2++) val -= A[c2 * 3 + col] * A[row * 3 + c2];
            if (row == col) {
                A[row * 3 + col] = val;
                inv_diag = 1 / val;
            } else {
                A[col * 3 + row] = val;
                A[row * 3 + col] = val * inv_diag;
            }
        }
    }
    for (int c = 0; c < 3; c++) {
        double y[3], x[3];
        for (int i = 0; i < 3; i++) {
            double val = (i == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) val -= A[i * 3 + j] * y[j];
            y[i] = val;
        }
        for (int i = 0; i < 3; i++) y[i] /= A[i * 3 + i];
        for (int i = 2; i >= 0; i--) {
            double val = y[i];
            for (int j = i + 1; j < 3; j++) val -= A[j * 3 + i] * x[j];
            x[i] = val;
        }
        for (int r = 0; r < 3; r++) out[r * 3 + c] = x[r];
    }
}

__global__ void __launch_bounds__(256) vinv_kernel(BaDev d, double lambda) {
    TL_MARK(d, 6)
    if (ba_guard_blocks(d)) return;
    vinv_body(d, lambda, blockIdx.x);
}
// The partial reduction and V*^-1 both only depend on K7 and not on each other: one launch, the first
// nx * RSPLIT workgroups reduce, the rest invert (a dependent launch costs ~2.3 us on this chip).
__global__ void __launch_bounds__(256) reduce_vinv_kernel(BaDev d, int grid_acc, int nx, double lambda, double lambda_stay) {
    TL_MARK(d, 5)
    if (ba_guard_blocks(d)) return;
    if (d.guard == 1 && d.sc->spec_stay) lambda = lambda_stay;
    const int b = blockIdx.x, nr = nx * RSPLIT;
    if (b < nr)
        reduce_partials_body(d, grid_acc, b % nx, b / nx);
    else
        vinv_body(d, lambda, b - nr);
}

// =================================================================================================
// K8: Schur complement, output-stationary camera-tile pairs
// =================================================================================================
#define SCHUR_TILE_ELEMS (SCHUR_TC * SCHUR_TC * 36 + SCHUR_TC * 6)   // 2304 + 48
#define SCHUR_BATCH 16                                                 // entries staged per LDS round

// Workgroup = (tile pair (a,b) of 8 x 8 cameras, a slice of the points touching both tiles), SOFTWARE
// PIPELINED: while the four waves multiply round i out of one LDS stage, the global loads of round i+1
// (16 entries x 16 loader lanes: the W blocks of the point's measurements in tile a / tile b, V*^-1,
// epsB) are in flight, and the work-list entries of round i+2 are being fetched.  After the product the
// loaders form Y = W V*^-1 and drop Y / W into the OTHER LDS stage; one barrier per round.
struct SchurPre {      // one loader lane's prefetched share of a round
    SchurEntry ent;
    int have;          // this lane's entry exists
    int m;             // measurement index (first iteration), -1 if none
    int f;             // its free-camera index
    double w[18];
    double vq;         // lane q < 9 of the entry carries Vinv[q]
    double eb;         // lane q < 3 carries epsB[q]
};

__device__ __forceinline__ void schur_fetch(const BaDev& d, const SchurEntry& ent, bool have, bool diag, int ls, SchurPre& p) {
    p.ent = ent;
    p.have = have;
    p.m = -1;
    p.f = -1;
    p.vq = 0;
    p.eb = 0;
    if (!have) return;
    const int na = ent.na_nb & 0xffff, nbm = (ent.na_nb >> 16) & 0xffff;
    const bool roleA = diag || ls < 8;
    const int l = roleA ? ls : ls - 8;
    const int n = roleA ? na : nbm;
    if (l < n) {
        p.m = (roleA ? ent.ma : ent.mb) + l;
        p.f = d.m_fidx[p.m];
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const double2 t = d.W[(size_t)q * d.M + p.m];
            p.w[2 * q] = t.x;
            p.w[2 * q + 1] = t.y;
        }
    }
    if (ls < 9) p.vq = d.Vinv[(size_t)ent.pt * 9 + ls];
    if (ls < 3) p.eb = d.epsB[(size_t)ent.pt * 3 + ls];
}

// ---- the product, on the matrix cores ---------------------------------------------------------------
// A round's 16 entries
// are laid out as two dense 48 x 48 operands  Y[3*entry + coord][6*slot + param]  and  W[...][...]
// (absent cameras: zero blocks), and the 48x48 partial tile is  Y W^T : nine 16x16 output tiles, the
// k dimension (point coordinates) taken four at a time by v_mfma_f64_16x16x4_f64
// (A[i = lane&15][k = lane>>4], B[k][j = lane&15]; D: column lane&15, row (lane>>4) + 4*v).  The four
// waves split the k-steps.  fp64 MFMA shares the vector FMA pipe on this chip (tools/pipes: the two do
// not overlap, and the MFMA sustains ~10 FMA/clk/SIMD against ~13 for v_fma_f64), so the gain is not
// flops: one MFMA replaces sixteen FMA instructions and their LDS operand traffic (6 ds_read_b64 per 9
// MFMAs instead of 18 ds_read_b128 per 108 FMAs).  A lane-per-block vector version of this product
// (36 accumulators per lane, operands re-read from LDS per entry) measured 99 us against 72 us.
#define SCH_K (3 * SCHUR_BATCH)   // k-values per round
#define SCH_ROWS (SCHUR_TC * 6)
#define SCH_LD (SCH_ROWS + 2)     // pitch of one k-row in doubles (even: a camera's six values are three 16-byte stores)
static_assert(SCH_ROWS == 48 && SCH_K % 4 == 0, "three 16-row fragments per operand");
struct SchurStageM {              // k-major: [3*entry + coord][6*slot + param] — a fragment reads 16 consecutive doubles
    double Y[SCH_K][SCH_LD];
    double W[SCH_K][SCH_LD];
    double eB[SCH_K];
};
typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned schur_put_m(SchurStageM& st, int le, bool roleA, bool diag, int slot, const double w[18],
                                                const double v[9]) {
    if (roleA) {
        double y[18];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) y[r * 3 + c] = w[r * 3] * v[c] + w[r * 3 + 1] * v[3 + c] + w[r * 3 + 2] * v[6 + c];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double2* dst = (double2*)&st.Y[3 * le + c][6 * slot];
#pragma unroll
            for (int h = 0; h < 3; h++) dst[h] = make_double2(y[(2 * h) * 3 + c], y[(2 * h + 1) * 3 + c]);
        }
    }
    if (!roleA || diag) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double2* dst = (double2*)&st.W[3 * le + c][6 * slot];
#pragma unroll
            for (int h = 0; h < 3; h++) dst[h] = make_double2(w[(2 * h) * 3 + c], w[(2 * h + 1) * 3 + c]);
        }
    }
    return 1u << slot;
}

// value of lane q of my 16-lane DPP row (row_newbcast), both halves of a double — VALU only, no LDS round trip
template <int Q>
__device__ __forceinline__ double row_bcast_f64(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x150 + Q, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + Q, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// OR over the 16 lanes of my DPP row (row rotations)
__device__ __forceinline__ unsigned row_or_u32(unsigned x) {
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true);   // row_ror:8
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, true);   // row_ror:4
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xf, 0xf, true);   // row_ror:2
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xf, 0xf, true);   // row_ror:1
    return x;
}

__device__ __forceinline__ void schur_store_m(const BaDev& d, SchurStageM& st, const SchurPre& p, bool diag, int a, int b, int le,
                                              int ls, int lane) {
    // V*^-1 of the entry: element q lives in lane q of the entry's 16 loader lanes (= one DPP row)
    const double v[9] = {row_bcast_f64<0>(p.vq), row_bcast_f64<1>(p.vq), row_bcast_f64<2>(p.vq),
                         row_bcast_f64<3>(p.vq), row_bcast_f64<4>(p.vq), row_bcast_f64<5>(p.vq),
                         row_bcast_f64<6>(p.vq), row_bcast_f64<7>(p.vq), row_bcast_f64<8>(p.vq)};
    const bool roleA = diag || ls < 8;
    unsigned bit = 0;
    if (ls < 3) st.eB[3 * le + ls] = p.have ? p.eb : 0.0;
    if (p.have) {
        if (p.m >= 0 && p.f >= 0) bit = schur_put_m(st, le, roleA, diag, p.f - (roleA ? a : b) * SCHUR_TC, p.w, v);
        // rare: more measurements in the tile range than loader lanes (fixed cameras interleaved)
        const int na = p.ent.na_nb & 0xffff, nbm = (p.ent.na_nb >> 16) & 0xffff;
        const int step = diag ? 16 : 8, n = roleA ? na : nbm;
        for (int l = (roleA ? ls : ls - 8) + step; l < n; l += step) {
            const int m = (roleA ? p.ent.ma : p.ent.mb) + l;
            const int f = d.m_fidx[m];
            if (f >= 0) {
                double w[18];
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    const double2 t = d.W[(size_t)q * d.M + m];
                    w[2 * q] = t.x;
                    w[2 * q + 1] = t.y;
                }
                bit |= schur_put_m(st, le, roleA, diag, f - (roleA ? a : b) * SCHUR_TC, w, v);
            }
        }
    }
    // presence of each camera slot: OR over the entry's loader lanes (a-role and b-role halves separately)
    const unsigned ba = row_or_u32(roleA ? bit : 0u), bb = row_or_u32(roleA ? 0u : bit);
    // absent slots (and whole entries past the end of the list) must read as zero blocks
    const int slot = ls & 7;
    const bool zy = ls < 8 && !((ba >> slot) & 1u);
    const bool zw = diag ? zy : (ls >= 8 && !((bb >> slot) & 1u));
    const double2 z2 = make_double2(0.0, 0.0);
    if (zy) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int h = 0; h < 3; h++) ((double2*)&st.Y[3 * le + c][6 * slot])[h] = z2;
    }
    if (zw) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int h = 0; h < 3; h++) ((double2*)&st.W[3 * le + c][6 * slot])[h] = z2;
    }
}

#ifdef K7_TIMING
#define SCH_T(acc, stmt) { const long long t0_ = (long long)__builtin_readcyclecounter(); stmt; acc += (long long)__builtin_readcyclecounter() - t0_; }
#else
#define SCH_T(acc, stmt) { stmt; }
#endif
__global__ void __launch_bounds__(256) schur_tile_mfma_kernel(BaDev d) {
    TL_MARK(d, 7)
    extern __shared__ __attribute__((aligned(16))) double schur_lds[];
    SchurStageM* stage = reinterpret_cast<SchurStageM*>(schur_lds);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const SchurWG wg = d.s_wgs[blockIdx.x];
    int a = (int)((sqrt(8.0 * wg.pair + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= wg.pair) a++;
    while (a * (a + 1) / 2 > wg.pair) a--;
    const int b = wg.pair - a * (a + 1) / 2;
    const bool diag = a == b;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int le = tid >> 4, ls = tid & 15;   // loader role: local entry, lane within the entry
    // 16-row fragments that hold cameras at all: the last tile of a system is usually partial (F = 50: two cameras =
    // one fragment of three), and its empty fragments are not multiplied
    const int rb_a = (6 * min(SCHUR_TC, d.F - a * SCHUR_TC) + 15) >> 4, rb_b = (6 * min(SCHUR_TC, d.F - b * SCHUR_TC) + 15) >> 4;
    v4f64 acc[3][3];
    double accE[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) acc[i][j] = (v4f64){0, 0, 0, 0};
    const int n_ent = wg.e_end - wg.e_begin;
    const int n_rounds = (n_ent + SCHUR_BATCH - 1) / SCHUR_BATCH;
    const SchurEntry none = {0, 0, 0, 0};
    auto entry_at = [&](int round, bool& have) {
        const int e = wg.e_begin + round * SCHUR_BATCH + le;
        have = round < n_rounds && e < wg.e_end;
        return have ? d.s_entries[e] : none;
    };
#ifdef K7_TIMING
    long long t_comp = 0, t_store = 0, t_fetch = 0, t_bar = 0, t_wait = 0;
    const long long t_start = (long long)__builtin_readcyclecounter();
#endif
    // prologue: round 0 into stage 0, entries of round 1 in registers
    SchurPre pre;
    bool have0, have_next;
    const SchurEntry e0 = entry_at(0, have0);
    schur_fetch(d, e0, have0, diag, ls, pre);
    SchurEntry ent_next = entry_at(1, have_next);
    schur_store_m(d, stage[0], pre, diag, a, b, le, ls, lane);
    __syncthreads();
    for (int i = 0; i < n_rounds; i++) {
        const bool more = i + 1 < n_rounds;
        SCH_T(t_fetch, if (more) schur_fetch(d, ent_next, have_next, diag, ls, pre));   // round i+1's data: loads in flight
        bool have2;
        const SchurEntry ent2 = entry_at(i + 2, have2);                  // round i+2's work-list entries
        // ---- round i on the matrix cores ----
        const SchurStageM& st = stage[i & 1];
        const int nb_ent = min(SCHUR_BATCH, n_ent - i * SCHUR_BATCH);
        const int n_ks = (3 * nb_ent + 3) >> 2;   // k-steps that hold data (the rest of the stage is zero)
#ifdef K7_TIMING
        const long long tc0 = (long long)__builtin_readcyclecounter();
#endif
        for (int s4 = wid; s4 < n_ks; s4 += 4) {
            const int kc = 4 * s4 + l4;
            double af[3], bf[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                af[t] = st.Y[kc][16 * t + l15];
                bf[t] = st.W[kc][16 * t + l15];
            }
#pragma unroll
            for (int ti = 0; ti < 3; ti++)
#pragma unroll
                for (int tj = 0; tj < 3; tj++)
                    if ((ti >= tj || !diag) && ti < rb_a && tj < rb_b)   // a diagonal pair only needs its lower triangle
                        acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ti], bf[tj], acc[ti][tj], 0, 0, 0);
            if (diag) {
                const double eb = st.eB[kc];
#pragma unroll
                for (int t = 0; t < 3; t++) accE[t] = fma(af[t], eb, accE[t]);
            }
        }
#ifdef K7_TIMING
        t_comp += (long long)__builtin_readcyclecounter() - tc0;
#endif
        // ---- stage round i+1 ----
#ifdef K7_TIMING
        SCH_T(t_wait, asm volatile("s_waitcnt vmcnt(0)" ::: "memory"));
#endif
        SCH_T(t_store, if (more) schur_store_m(d, stage[(i + 1) & 1], pre, diag, a, b, le, ls, lane));
        ent_next = ent2;
        have_next = have2;
        SCH_T(t_bar, __syncthreads());
    }
#ifdef K7_TIMING
    if (blockIdx.x == 100 && tid == 0) {
        d.dbg[10] = (long long)__builtin_readcyclecounter() - t_start;
        d.dbg[11] = t_fetch;
        d.dbg[12] = t_comp;
        d.dbg[13] = t_store;
        d.dbg[14] = t_bar;
        d.dbg[15] = n_rounds;
        d.dbg[9] = t_wait;
    }
#endif
    // E partials: sum the four k-quarters of a row (lanes l15 + 16 q)
#pragma unroll
    for (int t = 0; t < 3; t++) {
        accE[t] += __shfl_xor(accE[t], 16, 64);
        accE[t] += __shfl_xor(accE[t], 32, 64);
    }
    // stages dead: the buffer becomes the cross-wave reduction scratch
    double(*red)[64][40] = reinterpret_cast<double(*)[64][40]>(schur_lds);
    double flat[39];
#pragma unroll
    for (int ti = 0; ti < 3; ti++)
#pragma unroll
        for (int tj = 0; tj < 3; tj++)
#pragma unroll
            for (int v = 0; v < 4; v++) flat[(ti * 3 + tj) * 4 + v] = acc[ti][tj][v];
#pragma unroll
    for (int t = 0; t < 3; t++) flat[36 + t] = accE[t];
    // cross-wave reduction in fixed order: (w2 -> w0, w3 -> w1), then (w1 -> w0)
    if (wid >= 2) {
#pragma unroll
        for (int i = 0; i < 39; i++) red[wid - 2][lane][i] = flat[i];
    }
    __syncthreads();
    if (wid < 2) {
#pragma unroll
        for (int i = 0; i < 39; i++) flat[i] += red[wid][lane][i];
    }
    __syncthreads();
    if (wid == 1) {
#pragma unroll
        for (int i = 0; i < 39; i++) red[0][lane][i] = flat[i];
    }
    __syncthreads();
    if (wid == 0) {
        double* out = d.s_part + (size_t)blockIdx.x * SCHUR_TILE_ELEMS;
#pragma unroll
        for (int ti = 0; ti < 3; ti++)
#pragma unroll
            for (int tj = 0; tj < 3; tj++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int row = 16 * ti + l4 + 4 * v, col = 16 * tj + l15;
                    const int j = row / 6, r = row - 6 * j, k = col / 6, c = col - 6 * k;
                    const int i = (ti * 3 + tj) * 4 + v;
                    out[(size_t)(j * SCHUR_TC + k) * 36 + r * 6 + c] = flat[i] + red[0][lane][i];
                }
        if (l4 == 0) {
#pragma unroll
            for (int t = 0; t < 3; t++) out[SCHUR_TC * SCHUR_TC * 36 + 16 * t + l15] = flat[36 + t] + red[0][lane][36 + t];
        }
    }
}

// grid (tile pair, slice): fixed-order sum of the partial tiles, add U* / epsA, write S (lower) and E.
// Every rank adds its OWN partial U* and epsA ((1+lambda) diag(U) is linear, so the sharded partials
// fold into the one all-reduce); the padding identity is added on rank 0 only.
#define SRED_SLICES 10   // 9 x 256 tile elements + 1 slice for E and the padding rows
__global__ void __launch_bounds__(256) schur_reduce_kernel(BaDev d, double lambda, int pad_identity) {
    TL_MARK(d, 8)
    const int pair = blockIdx.x, slice = blockIdx.y;
    int a = (int)((sqrt(8.0 * pair + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= pair) a++;
    while (a * (a + 1) / 2 > pair) a--;
    const int b = pair - a * (a + 1) / 2;
    const int wg0 = d.s_pair_wg_begin[pair], wg1 = d.s_pair_wg_begin[pair + 1];
    double* S = d.SE;
    double* E = d.SE + (size_t)d.npad * d.npad;
    const int npad = d.npad;
    const size_t FS = (size_t)d.F * 27;
    if (slice < 9) {
        const int idx = slice * 256 + threadIdx.x;
        const int jk = idx / 36, rc = idx - jk * 36;
        const int j = jk >> 3, k = jk & 7, r = rc / 6, c = rc - r * 6;
        const int fa = a * SCHUR_TC + j, fb = b * SCHUR_TC + k;
        if (fa >= d.F || fb >= d.F) return;
        if (a == b && j < k) return;   // strict upper blocks are never read
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        const double* p = d.s_part + (size_t)wg0 * SCHUR_TILE_ELEMS + idx;
        int w = wg0;
        // (eight, then four partial tiles per round trip: the loop is a chain of dependent global round trips otherwise)
        for (; w + 8 <= wg1; w += 8, p += 8 * (size_t)SCHUR_TILE_ELEMS) {
            const double q0 = p[0], q1 = p[SCHUR_TILE_ELEMS], q2 = p[2 * (size_t)SCHUR_TILE_ELEMS], q3 = p[3 * (size_t)SCHUR_TILE_ELEMS];
            const double q4 = p[4 * (size_t)SCHUR_TILE_ELEMS], q5 = p[5 * (size_t)SCHUR_TILE_ELEMS], q6 = p[6 * (size_t)SCHUR_TILE_ELEMS],
                         q7 = p[7 * (size_t)SCHUR_TILE_ELEMS];
            s0 += q0 + q4;
            s1 += q1 + q5;
            s2 += q2 + q6;
            s3 += q3 + q7;
        }
        for (; w + 4 <= wg1; w += 4, p += 4 * (size_t)SCHUR_TILE_ELEMS) {
            s0 += p[0];
            s1 += p[SCHUR_TILE_ELEMS];
            s2 += p[2 * (size_t)SCHUR_TILE_ELEMS];
            s3 += p[3 * (size_t)SCHUR_TILE_ELEMS];
        }
        for (; w < wg1; w++, p += SCHUR_TILE_ELEMS) s0 += p[0];
        double val = -((s0 + s1) + (s2 + s3));
        if (a == b && j == k) {
            // U* : symmetrised U with diag * (1 + lambda)  (:383-390)
            const int rr = r >= c ? r : c, cc = r >= c ? c : r;
            double u = 0;
#pragma unroll
            for (int sp = 0; sp < RSPLIT; sp++) u += d.Usplit[sp * FS + fa * 27 + rr * (rr + 1) / 2 + cc];
            if (r == c) u *= (1.0 + lambda);
            val += u;
        }
        S[(size_t)(6 * fa + r) * npad + 6 * fb + c] = val;
        return;
    }
    if (a == b && threadIdx.x < SCHUR_TC * 6) {
        const int idx = threadIdx.x;
        const int j = idx / 6, r = idx - j * 6;
        const int fa = a * SCHUR_TC + j;
        if (fa < d.F) {
            double s = 0;
            for (int w = wg0; w < wg1; w++) s += d.s_part[(size_t)w * SCHUR_TILE_ELEMS + SCHUR_TC * SCHUR_TC * 36 + idx];
            double ea = 0;
#pragma unroll
            for (int sp = 0; sp < RSPLIT; sp++) ea += d.Usplit[sp * FS + fa * 27 + 21 + r];
            E[6 * fa + r] = ea - s;
        }
    }
    if (pair == 0) {
        // padding rows n..npad-1: identity so that the blocked factorisation is well defined
        const int np = d.npad - d.n;
        for (int idx = threadIdx.x; idx < np * d.npad; idx += 256) {
            const int r = d.n + idx / d.npad, c = idx % d.npad;
            if (c <= r) S[(size_t)r * npad + c] = (r == c && pad_identity) ? 1.0 : 0.0;
        }
        for (int idx = threadIdx.x; idx < np; idx += 256) E[d.n + idx] = 0.0;
    }
}

// =================================================================================================
// K10
// =================================================================================================
__global__ void __launch_bounds__(64) pose_update_kernel(BaDev d, int cur) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < d.C) {
        const double* T = d.pose[cur] + 12 * c;
        double* Tn = d.pose[cur ^ 1] + 12 * c;
        const int f = d.cam_free[c];
        if (f < 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) Tn[i] = T[i];
        } else {
            double mu[6], Tl[12], o[12];
#pragma unroll
            for (int i = 0; i < 6; i++) mu[i] = d.da[6 * f + i];
#pragma unroll
            for (int i = 0; i < 12; i++) Tl[i] = T[i];
            se3_exp_mul(mu, Tl, o);   // exp(da_j) * se3CfW  (:501)
#pragma unroll
            for (int i = 0; i < 12; i++) Tn[i] = o[i];
        }
    }
    if (blockIdx.x == 0) {
        double s = 0;
        for (int i = threadIdx.x; i < d.n; i += 64) s += d.da[i] * d.da[i];
        s = wave_sum_f64(s);
        if (threadIdx.x == 0) d.sc->sumsq_cam = s;
    }
}

// delta b, trial points, new robust error; block owns whole points
__global__ void __launch_bounds__(BA_CHUNK) point_update_kernel(DevCam cam, BaDev d, int cur, int est) {
    TL_MARK(d, 11)
    __shared__ double Ts[BA_CHUNK][3];
    __shared__ double Np[BA_CHUNK][3];
    __shared__ double wred[BA_CHUNK / 64][2];
    const int tid = threadIdx.x;
    const BaChunk ch = d.chunks[blockIdx.x];
    const int m = ch.m_begin + tid;
    const bool active = m < ch.m_end;
    // every per-measurement load leaves at once (clamped index, masked afterwards): W does not wait for the
    // state -> camera -> free-index chain, and the inputs of the new-error pass are here long before they are used
    const int mc = min(m, d.M - 1);
    double w[18];
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const double2 t = d.W[(size_t)q * d.M + mc];
        w[2 * q] = t.x;
        w[2 * q + 1] = t.y;
    }
    const int st = active ? (int)d.m_state[mc] : (int)MS_DEAD;
    const int c = d.m_cam[mc], p = d.m_pt[mc];
    const double2 fo = d.m_found[mc];
    const double sn = d.m_s[mc];
    const int f = d.cam_free[c];
    double t0 = 0, t1 = 0, t2 = 0;
    if (st == MS_ALIVE && f >= 0) {   // non-fixed, non-bad  (:469-474)
        const double* da = d.da + 6 * f;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const double a = da[r];
            t0 += w[r * 3] * a;
            t1 += w[r * 3 + 1] * a;
            t2 += w[r * 3 + 2] * a;
        }
    }
    Ts[tid][0] = t0;
    Ts[tid][1] = t1;
    Ts[tid][2] = t2;
    __syncthreads();
    // per point: eight threads add strided slices of its measurements' terms, the slices meet in a fixed order (DPP)
    const int npts = ch.pt_end - ch.pt_begin;
    double sq = 0;
    for (int base = 0; base < npts; base += BA_CHUNK / 8) {
        const int pi = base + (tid >> 3), sub = tid & 7;
        const bool on = pi < npts;
        const int pp = ch.pt_begin + (on ? pi : 0);
        const int r0 = d.rowptr[pp] - ch.m_begin, r1 = on ? d.rowptr[pp + 1] - ch.m_begin : r0;
        const double* eb = d.epsB + (size_t)pp * 3;
        const double* Vi = d.Vinv + (size_t)pp * 9;
        const double* X = d.pt[cur] + (size_t)pp * 3;
        const double e0 = eb[0], e1 = eb[1], e2 = eb[2];
        double vi[9], x[3];
#pragma unroll
        for (int i = 0; i < 9; i++) vi[i] = Vi[i];
#pragma unroll
        for (int i = 0; i < 3; i++) x[i] = X[i];
        double s0 = 0, s1 = 0, s2 = 0;
        for (int rr = r0 + sub; rr < r1; rr += 8) {
            s0 += Ts[rr][0];
            s1 += Ts[rr][1];
            s2 += Ts[rr][2];
        }
        s0 += dpp_row_shr_f64<1>(s0), s1 += dpp_row_shr_f64<1>(s1), s2 += dpp_row_shr_f64<1>(s2);
        s0 += dpp_row_shr_f64<2>(s0), s1 += dpp_row_shr_f64<2>(s1), s2 += dpp_row_shr_f64<2>(s2);
        s0 += dpp_row_shr_f64<4>(s0), s1 += dpp_row_shr_f64<4>(s1), s2 += dpp_row_shr_f64<4>(s2);
        if (on && sub == 7) {   // (lane 7 of the group holds the sum of its eight slices)
            const double v0 = e0 - s0, v1 = e1 - s1, v2 = e2 - s2;
            const double d0 = vi[0] * v0 + vi[1] * v1 + vi[2] * v2;
            const double d1 = vi[3] * v0 + vi[4] * v1 + vi[5] * v2;
            const double d2 = vi[6] * v0 + vi[7] * v1 + vi[8] * v2;
            sq += d0 * d0 + d1 * d1 + d2 * d2;
            const double n0 = x[0] + d0, n1 = x[1] + d1, n2 = x[2] + d2;   // :503-504
            double* Xn = d.pt[cur ^ 1] + (size_t)pp * 3;
            Xn[0] = n0;
            Xn[1] = n1;
            Xn[2] = n2;
            Np[pi][0] = n0;
            Np[pi][1] = n1;
            Np[pi][2] = n2;
        }
    }
    __syncthreads();
    double err = 0;
    if (active && st != MS_DEAD) {   // FindNewError over every listed measurement (:188-207)
        const double* T = d.pose[cur ^ 1] + 12 * c;
        const double* X = Np[p - ch.pt_begin];
        BaProj pr;
        if (!ba_project(cam, T, X, pr)) {
            err = 1.0;
            d.m_zbad_t[m] = 1;
        } else {
            const double ex = sn * (fo.x - pr.u), ey = sn * (fo.y - pr.v);
            const double s2 = d.sc->sigma_sq;
            const double e2n = ex * ex + ey * ey;
            err = ba_objective(est, e2n, s2, 1.0 / s2);
            // if this trial is accepted, these ARE pass 1's results of the next LM step (same poses,
            // points and projection code): keep them so that the next step can skip its projection pass
            d.m_e2t[m] = e2n;
            d.m_zbad_t[m] = 0;
        }
    }
    err = wave_sum_f64(err);
    sq = wave_sum_f64(sq);
    if ((tid & 63) == 0) {
        wred[tid >> 6][0] = err;
        wred[tid >> 6][1] = sq;
    }
    __syncthreads();
    if (tid == 0) {
        double e = 0, s = 0;
        for (int i = 0; i < BA_CHUNK / 64; i++) {
            e += wred[i][0];
            s += wred[i][1];
        }
        d.err_part[2 * blockIdx.x] = e;
        d.err_part[2 * blockIdx.x + 1] = s;
    }
}

__global__ void __launch_bounds__(256) finalize_new_kernel(BaDev d, double conv_limit, int last_allowed, ulonglong2* host_slots,
                                                            unsigned long long seq) {
    TL_MARK(d, 12)
    __shared__ double w[4][2];
    double e = 0, s = 0;
    // fixed assignment + fixed combine order: deterministic
    for (int i = threadIdx.x; i < d.n_chunks; i += 256) {
        e += d.err_part[2 * i];
        s += d.err_part[2 * i + 1];
    }
    e = wave_sum_f64(e);
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) {
        w[threadIdx.x >> 6][0] = e;
        w[threadIdx.x >> 6][1] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ne = w[0][0] + w[1][0] + w[2][0] + w[3][0], sp = w[0][1] + w[1][1] + w[2][1] + w[3][1];
        d.sc->new_err = ne;
        d.sc->sumsq_pt = sp;
        // the host's decision (src/Bundle.cc:338, :488-490, :518-533), taken here as well for the guarded kernels
        const double ce = d.sc->cur_err;
        const bool conv = d.sc->sumsq_cam + sp < conv_limit;
        const bool end_step = !(ne > ce) || conv || last_allowed != 0;
        d.sc->end_step = end_step ? 1 : 0;
        d.sc->spec_go = (end_step && ne < ce && !conv && last_allowed == 0) ? 1 : 0;
        d.sc->spec_stay = (end_step && !(ne < ce) && !conv && last_allowed == 0) ? 1 : 0;
    }
    // single-device runs publish the scalars from here (publish_scalars_kernel's job, one launch less per trial)
    if (host_slots) {
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x < sizeof(BaScalars) / 8)
            host_slots[threadIdx.x] = make_ulonglong2(((const volatile unsigned long long*)d.sc)[threadIdx.x], seq);
    }
}

// erase bad measurements, append to the outlier list (:536-547)
__global__ void __launch_bounds__(256) purge_kernel(BaDev d) {
    TL_MARK(d, 15)
    if (ba_guard_blocks(d)) return;
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool bad = m < d.M && d.m_state[m] == MS_BAD;
    const unsigned long long mask = __ballot(bad);
    if (mask == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&d.sc->n_outliers, __popcll(mask));   // one atomic per wave
    base = __shfl(base, 0, 64);
    if (bad) {
        d.outliers[base + __popcll(mask & ((1ull << lane) - 1ull))] = d.m_orig[m];
        d.m_state[m] = MS_DEAD;
    }
}

// The host's per-trial decision needs the scalars: they are written into host-mapped memory as
// (word, sequence) pairs, one 16-byte store per lane, and the host spins until every pair carries the
// expected sequence number — no system-scope fence on the device, no D2H copy + stream synchronise
// (~30 us of idle GPU per trial) on the host.
// Results leave the device through kernels too: poses, points and the outlier indices are stored straight into the
// context's host-mapped staging buffer, and a second (stream-ordered) launch stamps a sequence word the host spins on.
// The copy engine + hipStreamSynchronize pair this replaces slept on an interrupt and, about one Compute() in eight,
// took 7 ms instead of 70 us to wake up (PTAM_DEBUG_STALL: "compute total" against "compute loop").
__global__ void __launch_bounds__(256) readback_kernel(const double* __restrict__ pose, size_t n_pose, const double* __restrict__ pts,
                                                       size_t n_pts, const int* __restrict__ out, size_t n_out, double* __restrict__ h_pose,
                                                       double* __restrict__ h_pts, int* __restrict__ h_out) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = i0; i < n_pose; i += stride) h_pose[i] = pose[i];
    for (size_t i = i0; i < n_pts; i += stride) h_pts[i] = pts[i];
    for (size_t i = i0; i < n_out; i += stride) h_out[i] = out[i];
}
__global__ void stamp_kernel(volatile unsigned long long* slot, unsigned long long seq) { *slot = seq; }

#define MBOX_WORDS (sizeof(BaScalars) / 8)
static_assert(sizeof(BaScalars) % 8 == 0 && MBOX_WORDS <= 64, "published as 64-bit words by one wave");
__global__ void __launch_bounds__(64) publish_scalars_kernel(const BaScalars* sc, ulonglong2* host_slots, unsigned long long seq) {
    const unsigned i = threadIdx.x;
    if (i < MBOX_WORDS) host_slots[i] = make_ulonglong2(((const unsigned long long*)sc)[i], seq);
}

__global__ void set_scalars_kernel(BaDev d, double cur_err, int n_bad) {
    d.sc->cur_err = cur_err;
    d.sc->n_bad = n_bad;
}
__global__ void set_new_kernel(BaDev d, double new_err, double sumsq_pt) {
    d.sc->new_err = new_err;
    d.sc->sumsq_pt = sumsq_pt;
}
__global__ void pack2_kernel(const BaScalars* sc, double* out, int which) {
    if (which == 0) {
        out[0] = sc->cur_err;
        out[1] = (double)sc->n_bad;
    } else {
        out[0] = sc->new_err;
        out[1] = sc->sumsq_pt;
    }
}
__global__ void unpack2_kernel(BaScalars* sc, const double* in, int which) {
    if (which == 0) {
        sc->cur_err = in[0];
        sc->n_bad = (int)(in[1] + 0.5);
    } else {
        sc->new_err = in[0];
        sc->sumsq_pt = in[1];
    }
}
__global__ void place_keys_kernel(const double* __restrict__ src, int n, double* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// =================================================================================================
// host
// =================================================================================================
struct ptam_ba {
    ptam_ctx* ctx;
    ptam_ba_opts opts;
    // inputs (insertion order), copied at add_* time like the reference
    std::vector<double> cam_pose;
    std::vector<uint8_t> cam_fixed;
    std::vector<double> pts;
    std::vector<int> m_cam, m_pt;
    std::vector<double> m_found, m_s;
    std::vector<uint8_t> m_dead;
    // results
    bool converged = false;
    int accepted = 0;
    std::vector<ptam_ba_trial> trials;
    std::vector<std::pair<int, int>> outliers;   // (point, camera)
    std::vector<int> raw_out, raw_out_ends;      // measurement indices as purged + segment ends (per LM step), not yet digested
    int raw_out_base = 0, raw_out_done = 0;
    // device
    bool prepared = false;
    BaDev d;
    void* block = nullptr;
    size_t block_bytes = 0, block_cap = 0;
    bool e2_is_current = false;   // m_e2 / m_state hold pass 1 of the CURRENT poses and points (the last step accepted nothing)
    int band_local = 0;     // block bandwidth of S needed by THIS process' points (ba->d.band: the one in force)
    int cur = 0;
    size_t smem_acc = 0;
    bool use_wave = false;
    int per_wave = 1, extra_waves = 0;
    // host-mapped mailbox the device publishes BaScalars into (the LM loop's one host decision per trial)
    struct Mailbox {
        struct Slot {
            volatile unsigned long long v, seq;
        } slot[sizeof(BaScalars) / 8];
    };
    Mailbox* mbox = nullptr;       // host address
    Mailbox* mbox_dev = nullptr;   // device address of the same memory
    unsigned long long mbox_seq = 0;
    bool published_by_finalize = false;
    bool trial_is_current = false;   // the last trial was accepted: its new-error pass == pass 1 of the next step
    int k7_threads = BA_CHUNK;
    bool k7_loop = false;
    std::vector<int> sorted_orig;   // sorted position -> insertion index
    // gather buffers (sharded mode)
    double* d_gather = nullptr;
    size_t gather_cap = 0;
    double* d_xchg = nullptr;   // small exchange buffer (counts, scalars)
    double* d_sel = nullptr;    // sharded select: [4096 histogram words][world counts][world x XCAND_CAP keys][same again: list]
    int xcand_cap = XCAND_CAP_DEFAULT;
    bool cur_pending = false;   // sharded: (current error, bad count) are waiting behind S|E for the next all-reduce
    bool slow_select = false;   // sharded select fell back to gathering every key (after a select_overflow)
    // profiling
    bool prof = false;
    hipEvent_t ev[PTAM_K_COUNT][2];
    bool ev_ok = false;
    bool ev_used[PTAM_K_COUNT];
    double k_ms[PTAM_K_COUNT];
    int k_n[PTAM_K_COUNT];
    // communicator
    int rank = 0, world = 1;
    ptam_allreduce_f64_fn comm = nullptr;
    void* comm_user = nullptr;
};

static void ba_free_device(ptam_ba* ba) {
    if (ba->block) {
        // the block's kernels may still be queued: the next owner only touches it through the same stream
        void* drop = ctx_cache_give(ba->ctx->dev_cache, ba->block, ba->block_cap);
        if (drop) hipFree(drop);
    }
    if (ba->d_gather) hipFree(ba->d_gather);
    if (ba->d_xchg) hipFree(ba->d_xchg);
    if (ba->d_sel) hipFree(ba->d_sel);
    ba->d_sel = nullptr;
    ba->block = nullptr;
    ba->d_gather = nullptr;
    ba->d_xchg = nullptr;
    ba->gather_cap = 0;
    ba->prepared = false;
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

static void ba_finish_outliers(ptam_ba* ba);
static int ba_prepare_impl(ptam_ba* ba) {
    ptam_ctx* ctx = ba->ctx;
    ba_finish_outliers(ba);   // erased measurements of earlier Compute() calls leave the problem here
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    ba_free_device(ba);
    BaDev& d = ba->d;
    std::memset(&d, 0, sizeof d);
    const int C = (int)ba->cam_fixed.size(), P = (int)(ba->pts.size() / 3);
    const int Mall = (int)ba->m_cam.size();
    // free-camera indices in insertion order  (nStartRow, src/Bundle.cc:52-57)
    std::vector<int> cam_free(C, -1);
    int F = 0;
    for (int c = 0; c < C; c++)
        if (!ba->cam_fixed[c]) cam_free[c] = F++;
    // live measurements sorted point-major (point, camera); ties keep insertion order
    std::vector<int> order;
    order.reserve(Mall);
    for (int i = 0; i < Mall; i++)
        if (!ba->m_dead[i]) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
        if (ba->m_pt[x] != ba->m_pt[y]) return ba->m_pt[x] < ba->m_pt[y];
        return ba->m_cam[x] < ba->m_cam[y];
    });
    const int M = (int)order.size();
    for (int i = 1; i < M; i++)
        if (ba->m_pt[order[i]] == ba->m_pt[order[i - 1]] && ba->m_cam[order[i]] == ba->m_cam[order[i - 1]]) {
            ptam_set_error("duplicate measurement of point %d by camera %d", ba->m_pt[order[i]], ba->m_cam[order[i]]);
            return PTAM_E_ARG;
        }
    ba->sorted_orig = order;
    std::vector<int> rowptr(P + 1, 0);
    for (int i = 0; i < M; i++) rowptr[ba->m_pt[order[i]] + 1]++;
    for (int p = 0; p < P; p++) {
        if (rowptr[p + 1] > BA_CHUNK) {
            ptam_set_error("point %d has %d measurements; the limit is %d cameras per point", p, rowptr[p + 1], BA_CHUNK);
            return PTAM_E_LIMIT;
        }
        rowptr[p + 1] += rowptr[p];
    }
    // chunks: consecutive whole points, at most BA_CHUNK measurements
    std::vector<BaChunk> chunks;
    {
        int p = 0;
        while (p < P) {
            BaChunk ch;
            ch.pt_begin = p;
            ch.m_begin = rowptr[p];
            int cnt = 0, np = 0;
            while (p < P && cnt + (rowptr[p + 1] - rowptr[p]) <= BA_CHUNK && np < BA_CHUNK) {
                cnt += rowptr[p + 1] - rowptr[p];
                p++;
                np++;
            }
            ch.pt_end = p;
            ch.m_end = rowptr[p];
            chunks.push_back(ch);
        }
    }
    // wave chunks (K7 wave variant): consecutive whole points, at most 64 measurements
    std::vector<BaChunk> wchunks;
    int max_row = 0;
    for (int p = 0; p < P; p++) max_row = std::max(max_row, rowptr[p + 1] - rowptr[p]);
    if (max_row <= 64) {
        int p = 0;
        while (p < P) {
            BaChunk ch;
            ch.pt_begin = p;
            ch.m_begin = rowptr[p];
            int cnt = 0;
            while (p < P && cnt + (rowptr[p + 1] - rowptr[p]) <= 64) {
                cnt += rowptr[p + 1] - rowptr[p];
                p++;
            }
            ch.pt_end = p;
            ch.m_end = rowptr[p];
            wchunks.push_back(ch);
        }
    }
    // Schur work lists
    const int n_tiles = (F + SCHUR_TC - 1) / SCHUR_TC;
    const int n_pairs = n_tiles * (n_tiles + 1) / 2;
    std::vector<std::vector<SchurEntry>> per_pair(n_pairs);
    {
        std::vector<int> t_first(n_tiles), t_cnt(n_tiles), touched;
        for (int p = 0; p < P; p++) {
            touched.clear();
            // measurement sub-range of each tile inside the point's row (camera ids ascending ->
            // free indices ascending; fixed cameras may sit in between and are skipped in-kernel)
            int last_tile = -1;
            for (int i = rowptr[p]; i < rowptr[p + 1]; i++) {
                const int f = cam_free[ba->m_cam[order[i]]];
                if (f < 0) continue;
                const int t = f / SCHUR_TC;
                if (t != last_tile) {
                    touched.push_back(t);
                    t_first[t] = i;
                    last_tile = t;
                }
                t_cnt[t] = i - t_first[t] + 1;
            }
            for (size_t ia = 0; ia < touched.size(); ia++)
                for (size_t ib = 0; ib <= ia; ib++) {
                    const int a = touched[ia], b = touched[ib];
                    SchurEntry e;
                    e.pt = p;
                    e.ma = t_first[a];
                    e.mb = t_first[b];
                    e.na_nb = t_cnt[a] | (t_cnt[b] << 16);
                    per_pair[a * (a + 1) / 2 + b].push_back(e);
                }
        }
    }
    std::vector<SchurEntry> s_entries;
    std::vector<SchurWG> s_wgs;
    std::vector<int> pair_wg_begin(n_pairs + 1, 0);
    {
        size_t total = 0;
        for (auto& v : per_pair) total += v.size();
        // ~2 resident workgroups per CU x 256 CUs in ONE round; whole LDS batches per workgroup.  Entries are dealt out
        // evenly although pairs with a partial last tile multiply fewer fragments: weighting by MFMA count measured
        // SLOWER (103 vs 83 us) — the staging of an entry costs more than its fragments
        int per_wg = (int)std::min<size_t>(2048, std::max<size_t>(64, total / 500 + 1));
        per_wg = (per_wg + SCHUR_BATCH - 1) / SCHUR_BATCH * SCHUR_BATCH;
        for (int pr = 0; pr < n_pairs; pr++) {
            pair_wg_begin[pr] = (int)s_wgs.size();
            const int base = (int)s_entries.size();
            s_entries.insert(s_entries.end(), per_pair[pr].begin(), per_pair[pr].end());
            const int cnt = (int)per_pair[pr].size();
            for (int o = 0; o < cnt; o += per_wg) s_wgs.push_back(SchurWG{pr, base + o, base + std::min(cnt, o + per_wg)});
        }
        pair_wg_begin[n_pairs] = (int)s_wgs.size();
    }

    d.C = C;
    d.F = F;
    d.P = P;
    d.M = M;
    d.n = 6 * F;
    d.npad = ((d.n + SOLVE_NB - 1) / SOLVE_NB) * SOLVE_NB;
    d.n_chunks = (int)chunks.size();
    d.n_wchunks = (int)wchunks.size();
    ba->use_wave = !wchunks.empty();
    d.n_tiles = n_tiles;
    d.n_pairs = n_pairs;
    d.n_schur_wg = (int)s_wgs.size();
    d.n_schur_entries = (int)s_entries.size();
    // persistent grid of the accumulate kernel: bounded by LDS residency, 2 x 256 CUs by default
    ba->smem_acc = ((((size_t)F * 27 + 1) & ~(size_t)1) + (ba->use_wave ? (size_t)C * 12 + 2 : (size_t)BA_CHUNK * 8)) * sizeof(double);
    // wave variant, two shapes:
    //  - few chunks (every 64-measurement chunk can be resident at once: <= 24 waves per CU):
    //    straight-line kernel, ONE chunk per wave, 512-thread workgroups (64 VGPRs);
    //  - many chunks: persistent 256-thread workgroups looping over `per_wave` consecutive chunks,
    //    which amortises the LDS prologue and the camera-partial flush.
    const int n64_all = (M + 63) / 64;
    ba->k7_loop = ba->use_wave && n64_all > 256 * 24;
    if (const char* e = getenv("PTAM_K7_LOOP")) ba->k7_loop = ba->use_wave && atoi(e) != 0;   // shape sweeps (tools/k7_only.py)
    ba->k7_threads = !ba->use_wave ? BA_CHUNK : (ba->k7_loop ? 256 : 512);
    int n_cu = 256;
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
        n_cu = prop.multiProcessorCount;
    }
    auto k7_fn = [&](int threads) -> const void* {
        return !ba->use_wave ? (const void*)jac_accum_kernel : k7_wave_fn(threads, ba->k7_loop, ba->opts.estimator);
    };
    auto k7_occupancy = [&](int threads, int* per_cu) -> int {
        const void* k7 = k7_fn(threads);
        if (ba->smem_acc > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(k7, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ba->smem_acc);
            if (e != hipSuccess) return PTAM_E_HIP;
        }
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k7, threads, ba->smem_acc) == hipSuccess ? PTAM_OK : PTAM_E_HIP;
    };
    int per_cu = 0;
    if (int rc = k7_occupancy(ba->k7_threads, &per_cu)) return rc;
    if (ba->k7_loop) {
        // many cameras: the LDS partials (F*27 + C*12 doubles per workgroup) bound the workgroups per CU,
        // so the wider workgroup keeps more waves resident
        int per_cu512 = 0;
        if (int rc = k7_occupancy(512, &per_cu512)) return rc;
        if (per_cu512 * 8 > per_cu * 4) {
            ba->k7_threads = 512;
            per_cu = per_cu512;
        }
    }
    per_cu = std::max(1, std::min(per_cu, 8));
    if (ba->use_wave && ba->k7_loop) {
        if (const char* e = getenv("PTAM_K7_THREADS")) {
            const int t = atoi(e);
            if (t == 256 || t == 512) {
                ba->k7_threads = t;
                if (int rc = k7_occupancy(t, &per_cu)) return rc;
                per_cu = std::max(1, std::min(per_cu, 8));
            }
        }
        if (const char* e = getenv("PTAM_K7_WG_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e)));
    }
    if (ba->use_wave) {
        // every wave gets the same number of consecutive 64-measurement chunks
        // one 64-measurement chunk per wave (straight-line kernel body: 68 VGPRs instead of ~160 for the
        // looping form, i.e. every chunk of a 250 k-measurement problem is resident at once)
        const int n64 = (M + 63) / 64;
        const int wpb = ba->k7_threads / 64;
        if (ba->k7_loop) {
            // every resident slot gets a workgroup; chunks are dealt out evenly (q or q + 1 per wave)
            d.grid_acc = std::max(1, std::min(n_cu * per_cu, n64 / wpb));
            const int waves = d.grid_acc * wpb;
            ba->per_wave = n64 / waves;
            ba->extra_waves = n64 - ba->per_wave * waves;
        } else {
            ba->per_wave = 1;
            ba->extra_waves = 0;
            d.grid_acc = std::max(1, (n64 + wpb - 1) / wpb);
        }
    } else
        d.grid_acc = std::max(1, std::min(d.n_chunks, n_cu * per_cu));

    // ---- carve one device allocation ------------------------------------------------------------
    Carver cv;
    const size_t Mz = std::max(M, 1), Pz = std::max(P, 1), Cz = std::max(C, 1), Fz = std::max(F, 1);
    const size_t o_pose0 = cv.take(Cz * 96), o_pose1 = cv.take(Cz * 96), o_camfree = cv.take(Cz * 4);
    const size_t o_pt0 = cv.take(Pz * 24), o_pt1 = cv.take(Pz * 24), o_V = cv.take(Pz * 48), o_epsB = cv.take(Pz * 24),
                 o_Vinv = cv.take(Pz * 72), o_rowptr = cv.take((Pz + 1) * 4),
                 o_cut = cv.take(ba->use_wave ? (size_t)((M + 63) / 64) * 2 * 72 : 8);
    const size_t o_mcam = cv.take(Mz * 4), o_mpt = cv.take(Mz * 4), o_mfound = cv.take(Mz * 16), o_ms = cv.take(Mz * 8),
                 o_morig = cv.take(Mz * 4), o_mfidx = cv.take(Mz * 4), o_mstate = cv.take(Mz), o_me2 = cv.take(Mz * 8), o_me2t = cv.take(Mz * 8), o_zbad = cv.take(Mz), o_W = cv.take(Mz * 144);
    const size_t o_U = cv.take(Fz * 27 * 8 * 16), o_Upart = cv.take((size_t)d.grid_acc * Fz * 27 * 8);
    const size_t n_part = std::max(d.n_chunks, d.grid_acc);
    const size_t o_errp = cv.take(n_part * 16 + 16), o_badp = cv.take((size_t)d.grid_acc * 4 + 16);
    const size_t o_chunks = cv.take(std::max<size_t>(1, chunks.size()) * sizeof(BaChunk));
    const size_t o_wchunks = cv.take(std::max<size_t>(1, wchunks.size()) * sizeof(BaChunk));
    const size_t o_hist = cv.take(2 * HIST_BINS * 4), o_cand = cv.take(Mz * 8);
    const size_t o_sent = cv.take(std::max<size_t>(1, s_entries.size()) * sizeof(SchurEntry)),
                 o_swg = cv.take(std::max<size_t>(1, s_wgs.size()) * sizeof(SchurWG)),
                 o_spw = cv.take((size_t)(n_pairs + 1) * 4),
                 o_spart = cv.take(std::max<size_t>(1, s_wgs.size()) * SCHUR_TILE_ELEMS * 8);
    const size_t npad = std::max(d.npad, SOLVE_NB);
    const size_t o_SE = cv.take((npad * npad + npad + 2) * 8), o_L = cv.take(npad * npad * 8), o_Dg = cv.take(npad * 8),
                 o_y = cv.take(npad * 8), o_da = cv.take(npad * 8);
    const size_t o_out = cv.take(Mz * 4), o_sc = cv.take(sizeof(BaScalars)), o_dbg = cv.take(65536);
    ba->block_bytes = cv.off;
    if (!ctx_cache_take(ctx->dev_cache, ba->block_bytes, &ba->block, &ba->block_cap)) {   // (a released bundle's block, if it fits)
        HIP_TRY(hipMalloc(&ba->block, ba->block_bytes));
        ba->block_cap = ba->block_bytes;
    }
    HIP_TRY(hipMemsetAsync(ba->block, 0, ba->block_bytes, ctx->stream));
    char* base = (char*)ba->block;
    d.pose[0] = (double*)(base + o_pose0);
    d.pose[1] = (double*)(base + o_pose1);
    d.cam_free = (int*)(base + o_camfree);
    d.pt[0] = (double*)(base + o_pt0);
    d.pt[1] = (double*)(base + o_pt1);
    d.V = (double*)(base + o_V);
    d.epsB = (double*)(base + o_epsB);
    d.Vinv = (double*)(base + o_Vinv);
    d.cut = ba->use_wave ? (double*)(base + o_cut) : nullptr;
    d.rowptr = (int*)(base + o_rowptr);
    d.m_cam = (int*)(base + o_mcam);
    d.m_pt = (int*)(base + o_mpt);
    d.m_found = (double2*)(base + o_mfound);
    d.m_s = (double*)(base + o_ms);
    d.m_orig = (int*)(base + o_morig);
    d.m_fidx = (int*)(base + o_mfidx);
    d.m_state = (uint8_t*)(base + o_mstate);
    d.m_e2 = (double*)(base + o_me2);
    d.m_e2t = (double*)(base + o_me2t);
    d.m_zbad_t = (uint8_t*)(base + o_zbad);
    d.W = (double2*)(base + o_W);
    d.Usplit = (double*)(base + o_U);
    d.Upart = (double*)(base + o_Upart);
    d.err_part = (double*)(base + o_errp);
    d.bad_part = (int*)(base + o_badp);
    d.chunks = (BaChunk*)(base + o_chunks);
    d.wchunks = (BaChunk*)(base + o_wchunks);
    d.hist = (unsigned*)(base + o_hist);
    d.cand = (double*)(base + o_cand);
    d.s_entries = (SchurEntry*)(base + o_sent);
    d.s_wgs = (SchurWG*)(base + o_swg);
    d.s_pair_wg_begin = (int*)(base + o_spw);
    d.s_part = (double*)(base + o_spart);
    d.SE = (double*)(base + o_SE);
    d.L = (double*)(base + o_L);
    d.Dg = (double*)(base + o_Dg);
    d.y = (double*)(base + o_y);
    d.da = (double*)(base + o_da);
    d.outliers = (int*)(base + o_out);
    d.sc = (BaScalars*)(base + o_sc);
    d.dbg = (long long*)(base + o_dbg);

    // ---- upload -------------------------------------------------------------------------------------
    std::vector<int> h_cam(Mz), h_pt(Mz), h_orig(Mz), h_fidx(Mz);
    std::vector<double> h_found(2 * Mz), h_s(Mz);
    for (int i = 0; i < M; i++) {
        const int o = order[i];
        h_cam[i] = ba->m_cam[o];
        h_pt[i] = ba->m_pt[o];
        h_orig[i] = o;
        h_fidx[i] = cam_free[ba->m_cam[o]];
        h_found[2 * i] = ba->m_found[2 * o];
        h_found[2 * i + 1] = ba->m_found[2 * o + 1];
        h_s[i] = ba->m_s[o];
    }
    // block bandwidth of the camera system: S_jk != 0 only if cameras j, k share a point.  Keyframes that see the same
    // points are neighbours in time (src/MapMaker.cc adds them in order), so for a long trajectory S is banded and the
    // blocked LDL^T never leaves the band (no pivoting -> no fill outside it).  A sharded bundle only knows its own
    // points: it keeps the full width (the other ranks' points may couple other cameras).
    {
        int band = 0;
        for (int p = 0; p < P; p++) {
            int lo = INT_MAX, hi = -1;
            for (int i = rowptr[p]; i < rowptr[p + 1]; i++)
                if (h_fidx[i] >= 0) {
                    lo = std::min(lo, h_fidx[i]);
                    hi = std::max(hi, h_fidx[i]);
                }
            if (hi >= 0) band = std::max(band, (6 * hi + 5) / SOLVE_NB - (6 * lo) / SOLVE_NB);
        }
        d.band = ba->band_local = band;
    }
#define UP(dst, src, bytes)                                                                        \
    if ((bytes) > 0) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream))
    UP(d.pose[0], ba->cam_pose.data(), (size_t)C * 96);
    UP(d.cam_free, cam_free.data(), (size_t)C * 4);
    UP(d.pt[0], ba->pts.data(), (size_t)P * 24);
    UP(d.rowptr, rowptr.data(), (size_t)(P + 1) * 4);
    UP(d.m_cam, h_cam.data(), (size_t)M * 4);
    UP(d.m_pt, h_pt.data(), (size_t)M * 4);
    UP(d.m_found, h_found.data(), (size_t)M * 16);
    UP(d.m_s, h_s.data(), (size_t)M * 8);
    UP(d.m_orig, h_orig.data(), (size_t)M * 4);
    UP(d.m_fidx, h_fidx.data(), (size_t)M * 4);
    UP(d.chunks, chunks.data(), chunks.size() * sizeof(BaChunk));
    UP(d.wchunks, wchunks.data(), wchunks.size() * sizeof(BaChunk));
    UP(d.s_entries, s_entries.data(), s_entries.size() * sizeof(SchurEntry));
    UP(d.s_wgs, s_wgs.data(), s_wgs.size() * sizeof(SchurWG));
    UP(d.s_pair_wg_begin, pair_wg_begin.data(), pair_wg_begin.size() * 4);
#undef UP
    HIP_TRY(ptam_stream_wait(ctx->stream));   // host staging vectors die here
    {
        const int rc_s = ba_solve_init();
        if (rc_s) return rc_s;
    }
    HIP_TRY(hipFuncSetAttribute((const void*)schur_tile_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(2 * sizeof(SchurStageM))));
    HIP_TRY(hipMalloc((void**)&ba->d_xchg, 4096));

    ba->cur = 0;
    ba->prepared = true;
    return PTAM_OK;
}

// ---- profiling --------------------------------------------------------------------------------
static void prof_begin(ptam_ba* ba, int k) {
    if (!ba->prof) return;
    hipEventRecord(ba->ev[k][0], ba->ctx->stream);
}
static void prof_end(ptam_ba* ba, int k) {
    if (!ba->prof) return;
    hipEventRecord(ba->ev[k][1], ba->ctx->stream);
    ba->ev_used[k] = true;
}
static void prof_collect(ptam_ba* ba) {   // call after a stream sync
    if (!ba->prof) return;
    for (int k = 0; k < PTAM_K_COUNT; k++)
        if (ba->ev_used[k]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ba->ev[k][0], ba->ev[k][1]) == hipSuccess) {
                ba->k_ms[k] += ms;
                ba->k_n[k]++;
            }
            ba->ev_used[k] = false;
        }
}

static int ba_allreduce(ptam_ba* ba, double* dptr, size_t count) {
    if (!(ba->comm && ba->world > 1)) return PTAM_OK;
    const int rc = ba->comm(ba->comm_user, dptr, count, (void*)ba->ctx->stream);
    if (rc != 0) {
        ptam_set_error("all-reduce hook failed (%d)", rc);
        return PTAM_E_COMM;
    }
    return PTAM_OK;
}

// pass 1 + sigma^2
static int ba_pass1_sigma(ptam_ba* ba) {
    ptam_ctx* ctx = ba->ctx;
    BaDev& d = ba->d;
    const bool sharded = ba->comm && ba->world > 1;
    const double min_s2 = ba->opts.min_sigma * ba->opts.min_sigma;
    const int build_hist = (sharded && ba->slow_select) ? 0 : 1;   // (the gather-everything path histograms the gathered keys)
    prof_begin(ba, PTAM_K_PROJECT);
    if (ba->trial_is_current && d.M > 0)
        hipLaunchKernelGGL(pass1_from_trial_kernel, dim3(std::max(1, std::min((d.M + 256 * P1_U - 1) / (256 * P1_U), 512))), dim3(256), 0, ctx->stream, d,
                           build_hist);
    else if (ba->e2_is_current && d.M > 0) {
        if (build_hist)
            hipLaunchKernelGGL(pass1_keep_kernel, dim3(std::max(1, std::min((d.M + 256 * P1_U - 1) / (256 * P1_U), 512))), dim3(256), 0, ctx->stream, d);
    } else if (d.n_chunks > 0)
        hipLaunchKernelGGL(project_e2_kernel, dim3(std::min(d.n_chunks, 512)), dim3(BA_CHUNK), 0, ctx->stream, ctx->cam, d,
                           ba->cur, build_hist);
    prof_end(ba, PTAM_K_PROJECT);
    prof_begin(ba, PTAM_K_SELECT);
    if (!sharded) {
        hipLaunchKernelGGL(select_compact_kernel, dim3(std::max(1, std::min((d.M + 1023) / 1024, 256))), dim3(256), 0,
                           ctx->stream, d, (const double*)d.m_e2, (long long)d.M, (const uint8_t*)d.m_state);
    } else if (!ba->slow_select) {
        if (!ba->d_sel) {   // sized by the world the communicator was set for (ptam_ba_set_comm drops it on a change)
            if (const char* e = getenv("PTAM_XCAND_CAP")) ba->xcand_cap = std::max(1, std::min(atoi(e), 1 << 20));
            HIP_TRY(hipMalloc((void**)&ba->d_sel, (HIST_BINS + (size_t)ba->world * (1 + 2 * (size_t)ba->xcand_cap)) * sizeof(double)));
        }
        double* hx = ba->d_sel;                  // histogram exchange
        double* xc = ba->d_sel + HIST_BINS;      // candidate exchange
        const int cap = ba->xcand_cap;
        double* list = xc + (size_t)ba->world * (1 + (size_t)cap);
        const size_t n_xc = (size_t)ba->world * (1 + (size_t)cap);
        auto reduce_hist = [&](unsigned* h) -> int {
            hipLaunchKernelGGL(hist_to_f64_kernel, dim3(HIST_BINS / 256), dim3(256), 0, ctx->stream, (const unsigned*)h, hx, HIST_BINS);
            const int rc = ba_allreduce(ba, hx, HIST_BINS);
            if (rc) return rc;
            hipLaunchKernelGGL(f64_to_hist_kernel, dim3(HIST_BINS / 256), dim3(256), 0, ctx->stream, (const double*)hx, h, HIST_BINS);
            return PTAM_OK;
        };
        int rc = reduce_hist(d.hist);
        if (rc) return rc;
        hipLaunchKernelGGL(select_compact_kernel, dim3(std::max(1, std::min((d.M + 1023) / 1024, 256))), dim3(256), 0,
                           ctx->stream, d, (const double*)d.m_e2, (long long)d.M, (const uint8_t*)d.m_state);
        rc = reduce_hist(d.hist + HIST_BINS);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(xc, 0, n_xc * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(select_stage_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, xc, ba->rank, ba->world, cap);
        rc = ba_allreduce(ba, xc, n_xc);
        if (rc) return rc;
        hipLaunchKernelGGL(select_finish_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, (const double*)xc, list, ba->world, cap,
                           ba->opts.estimator, min_s2);
        prof_end(ba, PTAM_K_SELECT);
        HIP_TRY(hipGetLastError());
        return PTAM_OK;
    } else {
        // all-gather of the valid e^2 built from two all-reduces (counts, then a zero-padded vector)
        int* d_cnt = (int*)(ba->d_xchg + 256);
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 4, ctx->stream));
        hipLaunchKernelGGL(compact_valid_kernel, dim3(std::max(1, std::min((d.M + 255) / 256, 1024))), dim3(256), 0,
                           ctx->stream, d, d.cand, d_cnt);
        int n_local = 0;
        HIP_TRY(hipMemcpyAsync(&n_local, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ptam_stream_wait(ctx->stream));
        std::vector<double> counts(ba->world, 0.0);
        counts[ba->rank] = n_local;
        HIP_TRY(hipMemcpyAsync(ba->d_xchg, counts.data(), ba->world * 8, hipMemcpyHostToDevice, ctx->stream));
        int rc = ba_allreduce(ba, ba->d_xchg, ba->world);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(counts.data(), ba->d_xchg, ba->world * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ptam_stream_wait(ctx->stream));
        long long total = 0, off = 0;
        for (int r = 0; r < ba->world; r++) {
            if (r == ba->rank) off = total;
            total += (long long)(counts[r] + 0.5);
        }
        if ((size_t)total > ba->gather_cap) {
            if (ba->d_gather) HIP_TRY(hipFree(ba->d_gather));
            ba->gather_cap = (size_t)total + (size_t)total / 8 + 1024;
            HIP_TRY(hipMalloc((void**)&ba->d_gather, ba->gather_cap * 16));   // keys + candidate area
        }
        HIP_TRY(hipMemsetAsync(ba->d_gather, 0, (size_t)total * 8, ctx->stream));
        if (n_local > 0)
            hipLaunchKernelGGL(place_keys_kernel, dim3((n_local + 255) / 256), dim3(256), 0, ctx->stream,
                               (const double*)d.cand, n_local, ba->d_gather + off);
        rc = ba_allreduce(ba, ba->d_gather, (size_t)total);
        if (rc) return rc;
        BaDev dg = d;
        dg.cand = ba->d_gather + ba->gather_cap;   // candidates go behind the gathered keys
        if (total > 0)
            hipLaunchKernelGGL(hist_keys_kernel, dim3((int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 1024))),
                               dim3(256), 0, ctx->stream, (const double*)ba->d_gather, total, d.hist);
        hipLaunchKernelGGL(select_compact_kernel, dim3((int)std::max<long long>(1, std::min<long long>((total + 1023) / 1024, 256))),
                           dim3(256), 0, ctx->stream, dg, (const double*)ba->d_gather, total, (const uint8_t*)nullptr);
        hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, dg, ba->opts.estimator, min_s2);
        prof_end(ba, PTAM_K_SELECT);
        HIP_TRY(hipGetLastError());
        return PTAM_OK;
    }
    hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, ba->opts.estimator, min_s2);
    prof_end(ba, PTAM_K_SELECT);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

static void launch_k7(ptam_ba* ba, int guard = 0) {
    ptam_ctx* ctx = ba->ctx;
    BaDev d = ba->d;
    d.guard = guard;
    int cur = guard ? (ba->cur ^ 1) : ba->cur;   // a guarded launch belongs to the next step: the trial state is current there
    if (ba->use_wave) {
        int est = ba->opts.estimator;
        void* args[] = {&ctx->cam, &d, &cur, &est, &ba->per_wave, &ba->extra_waves};
        (void)hipLaunchKernel(k7_wave_fn(ba->k7_threads, ba->k7_loop, est), dim3(d.grid_acc), dim3(ba->k7_threads), args,
                              ba->smem_acc, ctx->stream);
    } else
        hipLaunchKernelGGL(jac_accum_kernel, dim3(d.grid_acc), dim3(BA_CHUNK), ba->smem_acc, ctx->stream, ctx->cam, d, cur,
                           ba->opts.estimator);
}

static int ba_pass2(ptam_ba* ba) {
    ptam_ctx* ctx = ba->ctx;
    BaDev& d = ba->d;
    prof_begin(ba, PTAM_K_JACOBIAN);
    launch_k7(ba);
    prof_end(ba, PTAM_K_JACOBIAN);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(std::max(1, (d.F * 27 + 63) / 64), RSPLIT), dim3(256), 0, ctx->stream,
                       d, d.grid_acc);
    HIP_TRY(hipGetLastError());
    if (ba->comm && ba->world > 1) {
        if (d.F > 0) {
            // current error / bad count ride behind S|E in the step's first camera-system all-reduce
            hipLaunchKernelGGL(pack2_kernel, dim3(1), dim3(1), 0, ctx->stream, (const BaScalars*)d.sc,
                               d.SE + (size_t)d.npad * d.npad + d.npad, 0);
            ba->cur_pending = true;
        } else {
            hipLaunchKernelGGL(pack2_kernel, dim3(1), dim3(1), 0, ctx->stream, (const BaScalars*)d.sc, ba->d_xchg, 0);
            int rc = ba_allreduce(ba, ba->d_xchg, 2);
            if (rc) return rc;
            hipLaunchKernelGGL(unpack2_kernel, dim3(1), dim3(1), 0, ctx->stream, d.sc, (const double*)ba->d_xchg, 0);
        }
    }
    return PTAM_OK;
}

static int ba_ensure_mailbox(ptam_ba* ba) {
    if (ba->mbox) return PTAM_OK;
    void* h = nullptr;
    size_t cap = 0;
    if (!ctx_cache_take(ba->ctx->host_cache, sizeof(ptam_ba::Mailbox), &h, &cap))
        HIP_TRY(hipHostMalloc(&h, sizeof(ptam_ba::Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h, 0, sizeof(ptam_ba::Mailbox));
    void* dv = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dv, h, 0));
    ba->mbox = (ptam_ba::Mailbox*)h;
    ba->mbox_dev = (ptam_ba::Mailbox*)dv;
    return PTAM_OK;
}

static int ba_trial(ptam_ba* ba, double lambda, bool skip_vinv, int last_allowed) {
    ptam_ctx* ctx = ba->ctx;
    BaDev& d = ba->d;
    prof_begin(ba, PTAM_K_VINV);
    if (d.P > 0 && !skip_vinv) hipLaunchKernelGGL(vinv_kernel, dim3((d.P + 255) / 256), dim3(256), 0, ctx->stream, d, lambda);
    prof_end(ba, PTAM_K_VINV);
    if (d.F > 0) {
        prof_begin(ba, PTAM_K_SCHUR);
        if (d.n_schur_wg > 0)
            hipLaunchKernelGGL(schur_tile_mfma_kernel, dim3(d.n_schur_wg), dim3(256), 2 * sizeof(SchurStageM), ctx->stream, d);
        hipLaunchKernelGGL(schur_reduce_kernel, dim3(d.n_pairs, SRED_SLICES), dim3(256), 0, ctx->stream, d, lambda,
                           (ba->world > 1 && ba->rank != 0) ? 0 : 1);
        prof_end(ba, PTAM_K_SCHUR);
        HIP_TRY(hipGetLastError());
        const size_t n_se = (size_t)d.npad * d.npad + d.npad;
        int rc = ba_allreduce(ba, d.SE, n_se + (ba->cur_pending ? 2 : 0));   // the path's one exchange step
        if (rc) return rc;
        if (ba->cur_pending) {
            hipLaunchKernelGGL(unpack2_kernel, dim3(1), dim3(1), 0, ctx->stream, d.sc, (const double*)(d.SE + n_se), 0);
            ba->cur_pending = false;
        }
        prof_begin(ba, PTAM_K_SOLVE);
        rc = ba_solve(ctx, d, ba->cur);
        prof_end(ba, PTAM_K_SOLVE);
        if (rc) return rc;
    }
    prof_begin(ba, PTAM_K_UPDATE);
    if (d.F == 0)   // no free camera: nothing was solved, the trial poses are copies
        hipLaunchKernelGGL(pose_update_kernel, dim3(std::max(1, (d.C + 63) / 64)), dim3(64), 0, ctx->stream, d, ba->cur);
    if (d.n_chunks > 0)
        hipLaunchKernelGGL(point_update_kernel, dim3(d.n_chunks), dim3(BA_CHUNK), 0, ctx->stream, ctx->cam, d, ba->cur,
                           ba->opts.estimator);
    {
        // single device, no per-kernel events: finalize publishes the scalars itself
        const bool fuse = !(ba->comm && ba->world > 1) && !ba->prof;
        ulonglong2* slots = nullptr;
        unsigned long long seq = 0;
        if (fuse) {
            const int rcm = ba_ensure_mailbox(ba);
            if (rcm) return rcm;
            slots = (ulonglong2*)ba->mbox_dev;
            seq = ++ba->mbox_seq;
            ba->published_by_finalize = true;
        }
        hipLaunchKernelGGL(finalize_new_kernel, dim3(1), dim3(256), 0, ctx->stream, d, ba->opts.update_sq_conv_limit, last_allowed,
                           slots, seq);
    }
    prof_end(ba, PTAM_K_UPDATE);
    HIP_TRY(hipGetLastError());
    if (ba->comm && ba->world > 1) {
        hipLaunchKernelGGL(pack2_kernel, dim3(1), dim3(1), 0, ctx->stream, (const BaScalars*)d.sc, ba->d_xchg, 1);
        int rc = ba_allreduce(ba, ba->d_xchg, 2);
        if (rc) return rc;
        hipLaunchKernelGGL(unpack2_kernel, dim3(1), dim3(1), 0, ctx->stream, d.sc, (const double*)ba->d_xchg, 1);
    }
    return PTAM_OK;
}

// The next LM step's prologue behind the device-side decision: purge (if the step ended), then — if the trial was
// accepted and the loop goes on — pass 1 from the trial's errors, the select, K7 with the trial state as current, the
// partial reduction and V*^-1 for lambda * 0.3.  Enqueued right after the scalars were published, i.e. while the host
// is still waiting for them.
static int ba_p1_blocks() {   // workgroup cap of the pass-1 kernels (each flushes its LDS histogram with global atomics)
    static const int n = [] {
        const char* e = getenv("PTAM_P1_BLOCKS");
        return e ? std::max(1, atoi(e)) : 512;
    }();
    return n;
}
static int ba_enqueue_speculative(ptam_ba* ba, double lambda_next, double lambda_stay) {
    ptam_ctx* ctx = ba->ctx;
    BaDev d = ba->d;
    const double min_s2 = ba->opts.min_sigma * ba->opts.min_sigma;
    d.guard = 1;
    hipLaunchKernelGGL(purge_pass1_kernel, dim3(std::max(1, std::min((d.M + 256 * P1_U - 1) / (256 * P1_U), ba_p1_blocks()))), dim3(256), 0, ctx->stream, d);
    hipLaunchKernelGGL(select_compact_kernel, dim3(std::max(1, std::min((d.M + 1023) / 1024, 256))), dim3(256), 0, ctx->stream, d,
                       (const double*)d.m_e2, (long long)d.M, (const uint8_t*)d.m_state);
    hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, ba->opts.estimator, min_s2);
    launch_k7(ba, 1);
    {
        const int nx = std::max(1, (d.F * 27 + 63) / 64);
        hipLaunchKernelGGL(reduce_vinv_kernel, dim3(nx * RSPLIT + (d.P + 255) / 256), dim3(256), 0, ctx->stream, d, d.grid_acc, nx,
                           lambda_next, lambda_stay);
    }
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

static int ba_publish_scalars(ptam_ba* ba);
static int ba_wait_scalars(ptam_ba* ba, BaScalars* out);
static int ba_read_scalars(ptam_ba* ba, BaScalars* out) {
    const int rc = ba_publish_scalars(ba);
    return rc ? rc : ba_wait_scalars(ba, out);
}

static int ba_publish_scalars(ptam_ba* ba) {
    ptam_ctx* ctx = ba->ctx;
    int rc = ba_ensure_mailbox(ba);
    if (rc) return rc;
    if (ba->published_by_finalize) {   // the trial's finalize kernel already wrote sequence number mbox_seq
        ba->published_by_finalize = false;
        return PTAM_OK;
    }
    const unsigned long long seq = ++ba->mbox_seq;
    hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(64), 0, ctx->stream, (const BaScalars*)ba->d.sc,
                       (ulonglong2*)ba->mbox_dev, seq);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

static int ba_wait_scalars(ptam_ba* ba, BaScalars* out) {
    ptam_ctx* ctx = ba->ctx;
    const unsigned long long seq = ba->mbox_seq;
    constexpr unsigned NW = sizeof(BaScalars) / 8;
    auto arrived = [&]() {
        for (unsigned i = 0; i < NW; i++)
            if (ba->mbox->slot[i].seq != seq) return false;
        return true;
    };
    if (ba->prof) {
        HIP_TRY(ptam_stream_wait(ctx->stream));   // the profiling events must have completed as well
    } else {
        // spin on the sequence words; every now and then ask the runtime whether the stream died instead
        unsigned spins = 0, idle_polls = 0, dbg_polls = 0;
        while (!arrived()) {
            if (++spins == 100000) {
                spins = 0;
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return PTAM_E_HIP;
                if (getenv("PTAM_DEBUG_WAIT") && (++dbg_polls % 100) == 0) {
                    std::fprintf(stderr, "[ptam] waiting for seq %llu: stream %s, slots", seq, q == hipSuccess ? "drained" : "busy");
                    for (unsigned i = 0; i < NW; i++) std::fprintf(stderr, " %llu", (unsigned long long)ba->mbox->slot[i].seq);
                    std::fprintf(stderr, "\n");
                }
                if (q == hipSuccess && !arrived() && ++idle_polls > 50) {
                    // the stream has drained and nothing published this sequence number: a logic error, not a wait
                    ptam_set_error("scalar mailbox: sequence %llu never arrived (slot 0 holds %llu)", seq,
                                   (unsigned long long)ba->mbox->slot[0].seq);
                    return PTAM_E_STATE;
                }
            }
        }
    }
    unsigned long long words[NW];
    for (;;) {
        std::atomic_thread_fence(std::memory_order_acquire);
        for (unsigned i = 0; i < NW; i++) words[i] = ba->mbox->slot[i].v;
        std::atomic_thread_fence(std::memory_order_acquire);
        if (arrived()) break;   // (cannot change again before the next publish; re-checked for torn 16-byte reads)
    }
    std::memcpy(out, words, sizeof(BaScalars));
    prof_collect(ba);
    return PTAM_OK;
}

// digest the purged-measurement indices of finished Compute() calls: per LM step sorted by insertion index, appended
// to the (point, camera) list, and the measurements marked erased for the next prepare
static void ba_finish_outliers(ptam_ba* ba) {
    int begin = ba->raw_out_done;
    for (size_t k = 0; k < ba->raw_out_ends.size(); k++) {
        const int end = ba->raw_out_ends[k];
        if (end <= begin) continue;
        std::sort(ba->raw_out.begin() + begin, ba->raw_out.begin() + end);
        for (int i = begin; i < end; i++) {
            const int o = ba->raw_out[i];
            ba->outliers.push_back(std::make_pair(ba->m_pt[o], ba->m_cam[o]));
            ba->m_dead[o] = 1;
        }
        begin = end;
    }
    ba->raw_out_done = begin;
    ba->raw_out_ends.clear();
}

extern "C" {

void ptam_ba_opts_default(ptam_ba_opts* o) {
    if (!o) return;
    o->max_iterations = 20;
    o->update_sq_conv_limit = 1e-6;
    o->min_sigma = 0.4;
    o->estimator = PTAM_EST_TUKEY;
    o->verbose = 0;
}

int ptam_ba_create(ptam_ctx* ctx, const ptam_ba_opts* opts, ptam_ba** out) {
    ARG_TRY(ctx && out);
    ptam_ba* ba = new ptam_ba();
    ba->ctx = ctx;
    if (opts)
        ba->opts = *opts;
    else
        ptam_ba_opts_default(&ba->opts);
    std::memset(&ba->d, 0, sizeof ba->d);
    std::memset(ba->k_ms, 0, sizeof ba->k_ms);
    std::memset(ba->k_n, 0, sizeof ba->k_n);
    std::memset(ba->ev_used, 0, sizeof ba->ev_used);
    *out = ba;
    return PTAM_OK;
}

int ptam_ba_destroy(ptam_ba* ba) {
    if (!ba) return PTAM_OK;
    hipSetDevice(ba->ctx->device);
    ptam_stream_wait(ba->ctx->stream);
    ba_free_device(ba);
    if (ba->mbox) {
        void* drop = ctx_cache_give(ba->ctx->host_cache, ba->mbox, sizeof(ptam_ba::Mailbox));
        if (drop) hipHostFree(drop);
    }
    if (ba->ev_ok)
        for (int k = 0; k < PTAM_K_COUNT; k++) {
            hipEventDestroy(ba->ev[k][0]);
            hipEventDestroy(ba->ev[k][1]);
        }
    delete ba;
    return PTAM_OK;
}

int ptam_ba_add_camera(ptam_ba* ba, const double pose[12], int fixed) {
    ARG_TRY(ba && pose);
    const int n = (int)ba->cam_fixed.size();
    ba->cam_pose.insert(ba->cam_pose.end(), pose, pose + 12);
    ba->cam_fixed.push_back(fixed ? 1 : 0);
    ba->prepared = false;
    return n;
}

int ptam_ba_add_point(ptam_ba* ba, const double pos[3]) {
    ARG_TRY(ba && pos);
    const int n = (int)(ba->pts.size() / 3);
    double v[3] = {pos[0], pos[1], pos[2]};
    if (std::isnan(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])) v[0] = v[1] = v[2] = 0;   // src/Bundle.cc:70-74
    ba->pts.insert(ba->pts.end(), v, v + 3);
    ba->prepared = false;
    return n;
}

int ptam_ba_add_meas(ptam_ba* ba, int cam, int point, const double found[2], double sigma_sq) {
    ARG_TRY(ba && found);
    ARG_TRY(cam >= 0 && cam < (int)ba->cam_fixed.size());
    ARG_TRY(point >= 0 && point < (int)(ba->pts.size() / 3));
    ba->m_cam.push_back(cam);
    ba->m_pt.push_back(point);
    ba->m_found.push_back(found[0]);
    ba->m_found.push_back(found[1]);
    ba->m_s.push_back(std::sqrt(1.0 / sigma_sq));   // dSqrtInvNoise src/Bundle.cc:91
    ba->m_dead.push_back(0);
    ba->prepared = false;
    return PTAM_OK;
}

int ptam_ba_add_cameras(ptam_ba* ba, int n, const double* poses12, const uint8_t* fixed) {
    ARG_TRY(ba && n >= 0 && (n == 0 || (poses12 && fixed)));
    for (int i = 0; i < n; i++) ptam_ba_add_camera(ba, poses12 + 12 * i, fixed[i]);
    return PTAM_OK;
}
int ptam_ba_add_points(ptam_ba* ba, int n, const double* pos3) {
    ARG_TRY(ba && n >= 0 && (n == 0 || pos3));
    for (int i = 0; i < n; i++) ptam_ba_add_point(ba, pos3 + 3 * i);
    return PTAM_OK;
}
int ptam_ba_add_measurements(ptam_ba* ba, int n, const int32_t* cam, const int32_t* point, const double* found2,
                             const double* sigma_sq) {
    ARG_TRY(ba && n >= 0 && (n == 0 || (cam && point && found2 && sigma_sq)));
    for (int i = 0; i < n; i++) {
        const int rc = ptam_ba_add_meas(ba, cam[i], point[i], found2 + 2 * i, sigma_sq[i]);
        if (rc < 0) return rc;
    }
    return PTAM_OK;
}

static int ba_ensure_mailbox(ptam_ba* ba);
int ptam_ba_prepare(ptam_ba* ba) {
    ARG_TRY(ba);
    if (ba->prepared) return PTAM_OK;
    int rc = ba_prepare_impl(ba);
    if (rc) return rc;
    // Every allocation Compute() would otherwise make on first use happens here: the mailbox and the pinned read-back
    // staging.  Mapping new host memory into the GPU's address space while kernels are queued makes the driver evict and
    // restore the queues — measured as a 10-28 ms wait for the first trial's scalars in about every third Compute().
    rc = ba_ensure_mailbox(ba);
    if (rc) return rc;
    void* pin = nullptr;
    return ctx_pinned(ba->ctx, 64 + (size_t)ba->d.C * 96 + (size_t)ba->d.P * 24 + (size_t)ba->d.M * 4 + 64, &pin);
}

int ptam_ba_set_profiling(ptam_ba* ba, int on) {
    ARG_TRY(ba);
    HIP_TRY(hipSetDevice(ba->ctx->device));
    if (on && !ba->ev_ok) {
        for (int k = 0; k < PTAM_K_COUNT; k++) {
            HIP_TRY(hipEventCreate(&ba->ev[k][0]));
            HIP_TRY(hipEventCreate(&ba->ev[k][1]));
        }
        ba->ev_ok = true;
    }
    ba->prof = on != 0;
    return PTAM_OK;
}

int ptam_ba_kernel_time(const ptam_ba* ba, int kernel, double* total_ms, int* launches) {
    ARG_TRY(ba && kernel >= 0 && kernel < PTAM_K_COUNT);
    if (total_ms) *total_ms = ba->k_ms[kernel];
    if (launches) *launches = ba->k_n[kernel];
    return PTAM_OK;
}

int ptam_ba_set_comm(ptam_ba* ba, int rank, int world, ptam_allreduce_f64_fn fn, void* user) {
    ARG_TRY(ba && world >= 1 && rank >= 0 && rank < world);
    ARG_TRY(world == 1 || fn);
    ARG_TRY(world <= 32);
    if (ba->d_sel && world != ba->world) {   // the select's exchange buffer is sized by the world
        HIP_TRY(hipSetDevice(ba->ctx->device));
        HIP_TRY(ptam_stream_wait(ba->ctx->stream));
        HIP_TRY(hipFree(ba->d_sel));
        ba->d_sel = nullptr;
    }
    ba->rank = rank;
    ba->world = world;
    ba->comm = fn;
    ba->comm_user = user;
    return PTAM_OK;
}

// Bundle::Compute src/Bundle.cc:116-158
#define BA_DBG(...) do { if (dbg_) { std::fprintf(stderr, "[ptam] " __VA_ARGS__); std::fprintf(stderr, "\n"); } } while (0)
int ptam_ba_compute(ptam_ba* ba, const volatile unsigned char* abort_flag, int* accepted_out) {
    ARG_TRY(ba);
    ptam_ctx* ctx = ba->ctx;
    const bool dbg_ = getenv("PTAM_DEBUG_WAIT") != nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    BA_DBG("compute: prepare");
    int rc = ptam_ba_prepare(ba);
    if (rc) return rc;
    BA_DBG("compute: prepared M=%d P=%d F=%d chunks=%d wchunks=%d grid_acc=%d schur_wg=%d", ba->d.M, ba->d.P, ba->d.F, ba->d.n_chunks,
           ba->d.n_wchunks, ba->d.grid_acc, ba->d.n_schur_wg);
    BaDev& d = ba->d;
    double lambda = 0.0001, lambda_factor = 2.0;   // :125-126
    ba->converged = false;
    ba->published_by_finalize = false;
    ba->cur_pending = false;
    ba->slow_select = false;
    ba->trial_is_current = false;
    ba->e2_is_current = false;
    bool hit_max = false;
    int counter = 0;
    ba->accepted = 0;
    ba->trials.clear();
    std::vector<int> step_outlier_end;   // outlier-list length after every LM step
    int n_steps = 0;
    bool prev_end_pending = false;   // the previous step's outlier-list length has not been read yet
    auto aborted = [&]() { return abort_flag && *abort_flag; };
    BaScalars sc;
    std::memset(&sc, 0, sizeof sc);
    double dbg_enq_ms = 0, dbg_wait_ms = 0;   // host time spent enqueueing trials / waiting for their scalars (PTAM_DEBUG_STALL)
    const auto dbg_t0 = std::chrono::steady_clock::now();
#ifdef K7_TIMING
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double ht0 = now_us();
    double ht_first = 0;
    (void)hipMemsetAsync(ba->d.dbg + TL_BASE, 0, 8, ba->ctx->stream);
#endif
    if (ba->comm && ba->world > 1) {
        // every rank must walk the same sequence of collectives: a rank without measurements would leave the loop below
        // at once and the others would wait for it for ever.  One tiny all-reduce up front makes the refusal unanimous.
        // The same exchange settles the block bandwidth of the camera system: every rank marks the bandwidth its own
        // points need (one-hot, the collective only sums), the widest one wins — S is the sum of all ranks' parts.
        const int nblk_x = d.npad / SOLVE_NB;
        const bool band_fits = nblk_x + 2 <= 512;   // d_xchg holds 512 doubles
        std::vector<double> mine(band_fits ? 2 + nblk_x : 2, 0.0), all(mine.size(), 0.0);
        mine[0] = d.M == 0 ? 1.0 : 0.0;
        mine[1] = (double)d.M;
        if (band_fits && nblk_x > 0) mine[2 + std::min(ba->band_local, nblk_x - 1)] = 1.0;
        HIP_TRY(hipMemcpyAsync(ba->d_xchg, mine.data(), mine.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        rc = ba_allreduce(ba, ba->d_xchg, mine.size());
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(all.data(), ba->d_xchg, all.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ptam_stream_wait(ctx->stream));
        ba->d.band = nblk_x;
        if (band_fits)
            for (int b = nblk_x - 1; b >= 0; b--)
                if (all[2 + b] > 0.5) {
                    ba->d.band = b;
                    break;
                }
        if (all[0] > 0.5) {
            ptam_set_error("sharded bundle: %d of %d ranks hold no measurement (shard the points so that every rank gets some)",
                           (int)(all[0] + 0.5), ba->world);
            return PTAM_E_STATE;
        }
    }
    const bool empty = d.M == 0;
    // speculative step prologue (ba_enqueue_speculative): single device, not while per-kernel events are being taken
    const bool spec = !(ba->comm && ba->world > 1) && !ba->prof && d.n_chunks > 0 && !getenv("PTAM_NO_SPECULATION");
    bool spec_ready = false;   // pass 1 .. V*^-1 of the coming step are already running behind the device-side flag
    while (!empty && !ba->converged && !hit_max && !aborted()) {
        // ---- Do_LM_Step :209-551 ----
        bool skip_vinv = false;
        if (spec_ready) {
            spec_ready = false;
            skip_vinv = true;
        } else {
            rc = ba_pass1_sigma(ba);
            if (rc) return rc;
            rc = ba_pass2(ba);
            if (rc) return rc;
        }
        bool have_cur = false;
        double cur_err = 0, new_err = 0;
        bool ran_any = false, redo_step = false, nan_stop = false;
        // while(dNewError > dCurrentError && !converged && !hitmax && !abort)  :338
        for (;;) {
            if (have_cur && !(new_err > cur_err)) break;
            if (ba->converged || hit_max || aborted()) break;
            BA_DBG("trial %d lambda %g", counter, lambda);
            const auto q0 = std::chrono::steady_clock::now();
            rc = ba_trial(ba, lambda, skip_vinv, counter + 1 >= ba->opts.max_iterations ? 1 : 0);
            {
                const double qms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - q0).count();
                dbg_enq_ms += qms;
                if (qms > 3.0 && getenv("PTAM_DEBUG_STALL"))
                    std::fprintf(stderr, "[ptam] stall: trial %d took %.2f ms to enqueue\n", counter, qms);
            }
            skip_vinv = false;
            if (rc) return rc;
            rc = ba_publish_scalars(ba);
            if (rc) return rc;
            if (spec) {   // what follows an accepted trial, queued while the host waits for the verdict
                rc = ba_enqueue_speculative(ba, lambda * 0.3, lambda);
                if (rc) return rc;
            }
            BA_DBG("trial %d enqueued, waiting", counter);
            {
                const auto w0 = std::chrono::steady_clock::now();
                rc = ba_wait_scalars(ba, &sc);
                const double wms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
                dbg_wait_ms += wms;
                if (wms > 3.0 && getenv("PTAM_DEBUG_STALL"))
                    std::fprintf(stderr, "[ptam] stall: trial %d waited %.2f ms for its scalars\n", counter, wms);
            }
            if (rc) return rc;
            BA_DBG("trial %d read: cur %g new %g sigma2 %g median %g n_valid %lld n_bad %d sumsq %g %g", counter, sc.cur_err, sc.new_err,
                   sc.sigma_sq, sc.median, sc.n_valid, sc.n_bad, sc.sumsq_cam, sc.sumsq_pt);
#ifdef K7_TIMING
            if (ht_first == 0) ht_first = now_us();
#endif
            if (!have_cur) {
                // (every step runs at least one trial: this first read of step s also carries the outlier-list
                //  length left by the purge that closed step s-1 — no separate read-back for it)
                if (prev_end_pending) step_outlier_end.push_back(sc.n_outliers);
                prev_end_pending = false;
                if (sc.select_overflow && !ba->slow_select) {
                    // sharded select: a rank's exchange slot overflowed (thousands of bit-identical errors), so this
                    // step ran with a wrong sigma^2.  Nothing has been committed: repeat it with the gather path.
                    ba->slow_select = true;
                    if (getenv("PTAM_DEBUG_SELECT")) std::fprintf(stderr, "[ptam] rank %d: select overflow, repeating the step on the gather path\n", ba->rank);
                    redo_step = true;
                    break;
                }
                have_cur = true;
                cur_err = sc.cur_err;
                new_err = cur_err + 9999;   // :337
                if (!(new_err > cur_err)) {
                    // NaN / inf current error: the reference never enters its trial loop and then calls Do_LM_Step again
                    // for ever (src/Bundle.cc:118-123 has no exit for it).  A library must not hang: give up here.
                    nan_stop = true;
                    break;
                }
            }
            ran_any = true;
            new_err = sc.new_err;
            const double sumsq = sc.sumsq_cam + sc.sumsq_pt;
            if (sumsq < ba->opts.update_sq_conv_limit) ba->converged = true;   // :488-490
            ptam_ba_trial t;
            t.lambda = lambda;
            t.sigma_sq = sc.sigma_sq;
            t.err_old = cur_err;
            t.err_new = new_err;
            t.sum_sq_update = sumsq;
            t.n_bad = sc.n_bad;
            t.accepted = 0;
            if (ba->opts.verbose)
                std::printf("L%.1e\tOld %.6f  New %.6f  Diff %.6f\n", lambda, cur_err, new_err, cur_err - new_err);
            if (new_err > cur_err) {   // ModifyLambda_BadStep :607-611
                lambda = lambda * lambda_factor;
                lambda_factor = lambda_factor * 2;
            }
            counter++;
            if (counter >= ba->opts.max_iterations) hit_max = true;   // :518-520
            ba->trials.push_back(t);
        }
        if (redo_step) continue;   // (keeps trial_is_current: pass 1 may still adopt the previous trial's errors)
        if (nan_stop) hit_max = true;
        ba->trial_is_current = false;
        ba->e2_is_current = ran_any && !(new_err < cur_err);   // nothing committed: the step's own pass 1 still describes the state
        if (ran_any && new_err < cur_err) {   // :523-533
            lambda_factor = 2.0;
            lambda *= 0.3;
            ba->cur ^= 1;   // commit: trial poses / points become current
            ba->accepted++;
            ba->trials.back().accepted = 1;
            ba->trial_is_current = true;
        }
        if (prev_end_pending) {   // a step that ran no trial (abort raised in between): read the length the slow way
            rc = ba_read_scalars(ba, &sc);
            if (rc) return rc;
            step_outlier_end.push_back(sc.n_outliers);
        }
        // the device took the same decision for the guarded kernels: accepted and the loop goes on -> they ran
        // (purge of this step included); otherwise they left at once and the step is closed here
        // ... or took the third way: the step ended with new == current error (nothing accepted, nothing rejected — the
        // update no longer moves a coordinate): the guarded kernels ran for the UNCHANGED state and lambda ("stay")
        const bool stayed = spec && ran_any && !(new_err < cur_err) && !(new_err > cur_err) && !ba->converged && !hit_max;
        spec_ready = (spec && ran_any && new_err < cur_err && !ba->converged && !hit_max) || stayed;
        if (!spec_ready && d.M > 0)
            hipLaunchKernelGGL(purge_kernel, dim3((d.M + 255) / 256), dim3(256), 0, ctx->stream, d);   // :536-547
        prev_end_pending = true;
        HIP_TRY(hipGetLastError());
        n_steps++;
    }
    BA_DBG("loop done, steps %d", n_steps);
    if (getenv("PTAM_DEBUG_STALL"))
        std::fprintf(stderr, "[ptam] compute loop: %zu trials in %.3f ms; host enqueueing trials %.3f ms, waiting for scalars %.3f ms\n",
                     ba->trials.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0).count(),
                     dbg_enq_ms, dbg_wait_ms);
    if (n_steps > 0) {   // the last step's purge
        rc = ba_read_scalars(ba, &sc);
        if (rc) return rc;
        step_outlier_end.push_back(sc.n_outliers);
    }
#ifdef K7_TIMING
    const double ht_loop = now_us();
#endif
#ifdef K7_TIMING
    {
        long long h[16];
        HIP_TRY(hipMemcpy(h, d.dbg, sizeof h, hipMemcpyDeviceToHost));
        std::printf("LDLT step2 wg0: loop %lld tail %lld | last wg (role %lld): loop %lld tail %lld\n", h[1] - h[0], h[2] - h[1], h[7],
                    h[5] - h[4], h[6] - h[5]);
        std::printf("SCHUR wg100 cycles: total %lld fetch %lld compute %lld wait-for-loads %lld store %lld barrier %lld rounds %lld\n", h[10], h[11],
                    h[12], h[9], h[13], h[14], h[15]);
    }
#endif
    // ---- read back results: one pinned staging buffer, one synchronisation (pageable destinations cost ~100 us each) ----
    const int n_out = step_outlier_end.empty() ? 0 : step_outlier_end.back();
    std::vector<int> out_idx(std::max(n_out, 1));
    {
        const size_t b_pose = (size_t)d.C * 96, b_pts = (size_t)d.P * 24, b_out = (size_t)n_out * 4;
        void* pin = nullptr;
        rc = ctx_pinned(ctx, 64 + b_pose + b_pts + b_out + 64, &pin);
        if (rc) return rc;
        char* hp = (char*)pin + 64;   // [sequence word | poses | points | outlier indices]
        char* dp = (char*)ctx->d_pinned + 64;
        volatile unsigned long long* slot = (volatile unsigned long long*)pin;
        const unsigned long long seq = ++ctx->pose_seq;
        *slot = 0;
        const size_t n_all = (size_t)d.C * 12 + (size_t)d.P * 3 + (size_t)n_out;
        hipLaunchKernelGGL(readback_kernel, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n_all + 255) / 256, 1024))), dim3(256), 0,
                           ctx->stream, (const double*)d.pose[ba->cur], (size_t)d.C * 12, (const double*)d.pt[ba->cur], (size_t)d.P * 3,
                           (const int*)d.outliers, (size_t)n_out, (double*)dp, (double*)(dp + b_pose), (int*)(dp + b_pose + b_pts));
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ctx->stream, (volatile unsigned long long*)ctx->d_pinned, seq);
        HIP_TRY(hipGetLastError());
        unsigned spins = 0;
        while (*slot != seq) {
            if (++spins == 100000) {
                spins = 0;
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return PTAM_E_HIP;
                if (q == hipSuccess && *slot != seq) return PTAM_E_HIP;   // drained without the stamp: something failed
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(ba->cam_pose.data(), hp, b_pose);
        if (d.P > 0) std::memcpy(ba->pts.data(), hp + b_pose, b_pts);
        if (n_out > 0) std::memcpy(out_idx.data(), hp + b_pose + b_pts, b_out);
    }
    // the outlier list (reference order: LM step, then insertion order) is put together when somebody asks for it
    // (ba_finish_outliers): sorting ~10 k indices costs as much as a whole lambda trial
    ba->raw_out.insert(ba->raw_out.end(), out_idx.begin(), out_idx.begin() + n_out);
    for (int end : step_outlier_end) ba->raw_out_ends.push_back(ba->raw_out_base + end);
    ba->raw_out_base += n_out;
    // the device copy stays valid for another Compute(): poses/points are current in pose[cur]; the
    // outlier counter restarts
    if (n_out > 0) {
        ba->prepared = false;   // rebuild without the erased measurements on the next Compute
    }
#ifdef K7_TIMING
    std::printf("HOST Compute: first trial read at %.1f us, loop end %.1f us, total %.1f us (%zu trials)\n", ht_first - ht0, ht_loop - ht0,
                now_us() - ht0, ba->trials.size());
    if (getenv("PTAM_TIMELINE")) {
        std::vector<long long> tl(2 + 2 * TL_MAX);
        HIP_TRY(hipMemcpy(tl.data(), d.dbg + TL_BASE, tl.size() * 8, hipMemcpyDeviceToHost));
        const long long n_tl = std::min<long long>(tl[0], TL_MAX);
        static const char* nm[] = {"?", "purge_pass1", "select_compact", "select_final", "K7", "reduce_vinv", "vinv", "schur_tile", "schur_reduce",
                                   "ldlt_step0", "backward", "point_update", "finalize", "project_e2", "pass1_trial", "purge", "reduce_partials", "publish"};
        for (long long i = 0; i < n_tl; i++)
            std::printf("TL %4lld %-16s start %9.2f us  (+%.2f)\n", i, nm[tl[2 + 2 * i] < 18 ? tl[2 + 2 * i] : 0], (tl[3 + 2 * i] - tl[3]) * 0.01,
                        i ? (tl[3 + 2 * i] - tl[1 + 2 * i]) * 0.01 : 0.0);
    }
#endif
    if (getenv("PTAM_DEBUG_STALL"))
        std::fprintf(stderr, "[ptam] compute total %.3f ms\n",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0).count());
    if (accepted_out) *accepted_out = ba->accepted;
    return PTAM_OK;
}

int ptam_ba_converged(const ptam_ba* ba) { return ba && ba->converged ? 1 : 0; }

int ptam_ba_get_point(const ptam_ba* ba, int n, double pos[3]) {
    ARG_TRY(ba && pos && n >= 0 && n < (int)(ba->pts.size() / 3));   // vector::at() would throw
    std::memcpy(pos, &ba->pts[3 * (size_t)n], 24);
    return PTAM_OK;
}
int ptam_ba_get_camera(const ptam_ba* ba, int n, double pose[12]) {
    ARG_TRY(ba && pose && n >= 0 && n < (int)ba->cam_fixed.size());
    std::memcpy(pose, &ba->cam_pose[12 * (size_t)n], 96);
    return PTAM_OK;
}
int ptam_ba_get_all(const ptam_ba* ba, double* poses12, double* points3) {
    ARG_TRY(ba);
    if (poses12) std::memcpy(poses12, ba->cam_pose.data(), ba->cam_pose.size() * 8);
    if (points3) std::memcpy(points3, ba->pts.data(), ba->pts.size() * 8);
    return PTAM_OK;
}
int ptam_ba_get_outliers(const ptam_ba* ba, int32_t* pairs, int cap) {
    if (!ba) return PTAM_E_ARG;
    ba_finish_outliers(const_cast<ptam_ba*>(ba));
    const int n = (int)ba->outliers.size();
    for (int i = 0; i < n && i < cap && pairs; i++) {
        pairs[2 * i] = ba->outliers[i].first;
        pairs[2 * i + 1] = ba->outliers[i].second;
    }
    return n;
}
int ptam_ba_get_trials(const ptam_ba* ba, ptam_ba_trial* out, int cap) {
    if (!ba) return PTAM_E_ARG;
    const int n = (int)ba->trials.size();
    for (int i = 0; i < n && i < cap && out; i++) out[i] = ba->trials[i];
    return n;
}
int ptam_ba_counts(const ptam_ba* ba, int* n_cams, int* n_free, int* n_points, int* n_meas) {
    ARG_TRY(ba);
    int f = 0;
    for (uint8_t x : ba->cam_fixed) f += x ? 0 : 1;
    ba_finish_outliers(const_cast<ptam_ba*>(ba));
    int live = 0;
    for (uint8_t x : ba->m_dead) live += x ? 0 : 1;
    if (n_cams) *n_cams = (int)ba->cam_fixed.size();
    if (n_free) *n_free = f;
    if (n_points) *n_points = (int)(ba->pts.size() / 3);
    if (n_meas) *n_meas = live;
    return PTAM_OK;
}

// K7 alone, HIP-event timed over `reps` launches (bench.py roofline leg)
int ptam_ba_bench_jacobian(ptam_ba* ba, int reps, double* avg_ms, double* algorithmic_bytes) {
    ARG_TRY(ba && reps > 0);
    ptam_ctx* ctx = ba->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ptam_ba_prepare(ba);
    if (rc) return rc;
    BaDev& d = ba->d;
    ARG_TRY(d.M > 0);
    const bool prof = ba->prof;
    ba->prof = false;
    rc = ba_pass1_sigma(ba);
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++)
        launch_k7(ba);
    HIP_TRY(ptam_stream_wait(ctx->stream));
    // one event pair around `reps` back-to-back launches: the average is the kernel's steady-state
    // duration (an event pair around a single launch adds ~6 us of record / completion latency — an empty
    // kernel measures 6.3 us that way, tools/membw — and would not agree with rocprofv3's kernel trace)
    HIP_TRY(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < reps; i++) launch_k7(ba);
    HIP_TRY(hipEventRecord(e1, ctx->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms_total = 0;
    HIP_TRY(hipEventElapsedTime(&ms_total, e0, e1));
    const double total = ms_total;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    ba->prof = prof;
#ifdef K7_TIMING
    {
        long long h[16];
        HIP_TRY(hipMemcpy(h, d.dbg, sizeof h, hipMemcpyDeviceToHost));
        std::printf("K7 stamps (10 ns ticks since kernel-body start):");
        for (int i = 1; i < 10; i++) std::printf(" [%d] %lld", i, h[i] - h[0]);
        std::printf("\n");
        const int nb = std::min(ba->d.grid_acc, 2000);
        std::vector<long long> w(2 * nb);
        HIP_TRY(hipMemcpy(w.data(), d.dbg + 16, w.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = w[0];
        for (int b = 0; b < nb; b++) t0 = std::min(t0, w[2 * b]);
        std::vector<long long> st(nb), en(nb), du(nb);
        for (int b = 0; b < nb; b++) st[b] = w[2 * b] - t0, en[b] = w[2 * b + 1] - t0, du[b] = en[b] - st[b];
        std::sort(st.begin(), st.end());
        std::sort(en.begin(), en.end());
        std::sort(du.begin(), du.end());
        std::printf("K7 block 7: body starts %lld ticks after the block's first instruction\n", h[0] - w[14]);
        std::printf("K7 wall (10 ns ticks, %d blocks): start p0/p50/p90/p100 %lld %lld %lld %lld | end %lld %lld %lld %lld | dur %lld %lld %lld %lld\n",
                    nb, st[0], st[nb / 2], st[nb * 9 / 10], st[nb - 1], en[0], en[nb / 2], en[nb * 9 / 10], en[nb - 1], du[0], du[nb / 2],
                    du[nb * 9 / 10], du[nb - 1]);
    }
#endif
    if (avg_ms) *avg_ms = total / reps;
    if (algorithmic_bytes)   // 8 idx + 24 found/s + 1 state + 144 W per measurement,
                             // 96 B/camera pose read, 24 read + 72 write per point, 216 B/free camera
        *algorithmic_bytes = (double)d.M * (8 + 24 + 1 + 144) + (double)d.C * 96 + (double)d.P * (24 + 72) + (double)d.F * 216;
    return PTAM_OK;
}

}   // extern "C"

// every kernel of this file resolved once, when a context is created: the first launch of a kernel otherwise pays for
// loading the code object / resolving the function — 10-28 ms in the middle of the first frame or the first adjustment
void ba_preload_kernels() {
    ptam_preload((const void*)project_e2_kernel);
    ptam_preload((const void*)pass1_from_trial_kernel);
    ptam_preload((const void*)pass1_keep_kernel);
    ptam_preload((const void*)purge_pass1_kernel);
    ptam_preload((const void*)hist_keys_kernel);
    ptam_preload((const void*)select_compact_kernel);
    ptam_preload((const void*)select_final_kernel);
    ptam_preload((const void*)hist_to_f64_kernel);
    ptam_preload((const void*)f64_to_hist_kernel);
    ptam_preload((const void*)select_stage_kernel);
    ptam_preload((const void*)select_finish_kernel);
    ptam_preload((const void*)compact_valid_kernel);
    ptam_preload((const void*)jac_accum_kernel);
    ptam_preload((const void*)reduce_partials_kernel);
    ptam_preload((const void*)vinv_kernel);
    ptam_preload((const void*)reduce_vinv_kernel);
    ptam_preload((const void*)schur_tile_mfma_kernel);
    ptam_preload((const void*)schur_reduce_kernel);
    ptam_preload((const void*)pose_update_kernel);
    ptam_preload((const void*)point_update_kernel);
    ptam_preload((const void*)finalize_new_kernel);
    ptam_preload((const void*)purge_kernel);
    ptam_preload((const void*)readback_kernel);
    ptam_preload((const void*)stamp_kernel);
    ptam_preload((const void*)publish_scalars_kernel);
    ptam_preload((const void*)set_scalars_kernel);
    ptam_preload((const void*)set_new_kernel);
    ptam_preload((const void*)pack2_kernel);
    ptam_preload((const void*)unpack2_kernel);
    ptam_preload((const void*)place_keys_kernel);
    for (int est = 0; est < 2; est++) {
        ptam_preload(k7_wave_fn(512, false, est ? PTAM_EST_CAUCHY : PTAM_EST_TUKEY));
        ptam_preload(k7_wave_fn(256, true, est ? PTAM_EST_CAUCHY : PTAM_EST_TUKEY));
        ptam_preload(k7_wave_fn(512, true, est ? PTAM_EST_CAUCHY : PTAM_EST_TUKEY));
    }
}
