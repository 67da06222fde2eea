// pose_device.h — the register-resident pose Gauss-Newton loop (n <= 1024 measurements, ONE workgroup) as a device template:
// pose.hip wraps it in the kernels behind ptam_pose_gn*, trackmap.hip in the fused kernels of the TrackMap chain, whose
// prologue takes the measurements straight from the search stage's slots (no gather launch in between).
#pragma once
#include "common.h"
#include "track_internal.h"

#define GS_LIMIT 1024   // measurements the register-resident kernel (pose_gn_small_kernel) holds

// TooN Cholesky<6> (unpivoted LDL^T, lower triangle) + backsub, run by one thread
// (reciprocals by v_rcp_f64 + two Newton steps, as everywhere in the bundle kernels: one thread runs this on the critical
//  path of every iteration, and an IEEE division is a dependent chain of ~12 instructions — twelve of them were half of it)
// RIGHT-LOOKING (round 5): after pivot k every element of the trailing triangle takes its update at once — the same
// subtractions in the same order per element as TooN's left-looking loops (pivots ascending, one FMA each: the same bits),
// but fifteen independent FMAs per pivot instead of chains of them: one lane's dependent fp64 instructions cost ~8 cycles
// each, independent ones 4.  The forward substitution is column-oriented for the same reason (same order: columns
// ascending); the backward one keeps TooN's row-wise order (columns ascending inside a row), which is its dependent form.
static __device__ void ldlt6_solve(double A[36], const double b[6], double x[6]) {
    double inv_d[6], l[6][6];   // l[i][k]: multipliers (i > k)
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double inv = rcp_nr(A[k * 6 + k]);
        inv_d[k] = inv;
        double u[6];
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
            u[i] = A[i * 6 + k];        // the undivided column
            l[i][k] = u[i] * inv;
        }
#pragma unroll
        for (int j = k + 1; j < 6; j++)
#pragma unroll
            for (int i = j; i < 6; i++) A[i * 6 + j] = fma(-u[j], l[i][k], A[i * 6 + j]);
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] = b[i];
#pragma unroll
    for (int j = 0; j < 5; j++)
#pragma unroll
        for (int i = j + 1; i < 6; i++) y[i] = fma(-l[i][j], y[j], y[i]);
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] *= inv_d[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double val = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; j++) val = fma(-l[j][i], x[j], val);
        x[i] = val;
    }
}

// entry pose handed over by value (kernel argument): a tracked frame's prediction comes from the host's motion model,
// 96 bytes that need no copy of their own
struct PoseIn {
    double v[12];
    int use;
};

// ------------------------------------------------------------------------------------------------
// Fast path, n <= 1024: ONE measurement per thread, all per-measurement state in registers, the e^2
// keys of the order statistic in LDS, wave sums by DPP (no LDS round trips), same arithmetic and
// the same fixed reduction order as the general kernel above.
// ------------------------------------------------------------------------------------------------
// The workgroup of the large instantiations and the measurements per thread of the largest one (n <= 1024).  Round 4: 512 x 2 —
// two waves per SIMD.  Rounds 2-3 ran 256 x 4 (one wave per SIMD, the four measurements of a thread interleaved as the only
// latency hiding) and had measured 512 x 2 as no faster; with the five-barrier order statistic and the wave-0-only finisher gone
// it is: every phase of an iteration is a chain of dependent instructions at ~6-8 cycles each, and a second wave per SIMD
// fills the gaps (frame 134.8 -> 129.2 us, A/B on one box: tools/dev/frame_ab.sh).
#ifndef GS_THREADS
#define GS_THREADS 512
#define GS_MPT 2
#endif
static_assert(GS_THREADS * GS_MPT == GS_LIMIT, "GS_LIMIT");
#define GS_WAVE_LIMIT 64 // lists of at most 64 measurements (the coarse set: Tracker.CoarseMax = 60) run as ONE wave — see pose_gn_small_kernel
#define GS_BINS 2048   // 11-bit digits of the order-statistic select

// THREADS x MPT measurements; THREADS = 256, or 64 (MPT = 1): a single wave, whose barriers cost nothing
template <int THREADS, int MPT>
struct GnSmallShared {
    static constexpr int WAVES = THREADS / 64;
    static constexpr int SLICES = THREADS / 32;            // 32-thread slices of the workgroup
    static constexpr int TR_PITCH = SLICES * 33 + 1;       // padded so that slices and rows fall into different banks
    double pose[12];
    double mu[6];
    double red[WAVES][27];
    double keys[THREADS * MPT];
    unsigned hist[GS_BINS];
    int sel_digit, sel_k, sel_cnt;
    int wcount[WAVES];
    int scan[WAVES];
    unsigned long long cand[64];
    int n_cand;
    // round 4, the two-barrier select (small_select_fast): a histogram pair used alternately — lane L's 32 bins at word j * 64 + L —
    // and a candidate list pair
    unsigned hist2[2][GS_BINS];
    unsigned long long cand2[2][64];
    int n_cand2[2];
    int book[20];   // the frame's integer results, parked until the result block is written (loader's begin -> publish)
    double tr[27][TR_PITCH];   // transposed per-thread partials of the 27 sums (row = sum, column = thread)
};

// wave sum by DPP: row shifts 1,2,4,8 then row broadcasts; the total lands in lane 63
__device__ __forceinline__ double wave_sum_f64_dpp(double v) {
    v += dpp_row_shr_f64<1>(v);
    v += dpp_row_shr_f64<2>(v);
    v += dpp_row_shr_f64<4>(v);
    v += dpp_row_shr_f64<8>(v);
    v += dpp_bcast_f64<0x142, 0xa>(v);
    v += dpp_bcast_f64<0x143, 0xc>(v);
    return v;
}

// exact k-th smallest of sh.keys[0..n) (non-found entries hold +inf), bit patterns compared as unsigned
// 64-bit integers.
//  - fast path: ONE histogram over the leading bits (sign, exponent, GS_KEY_MBITS mantissa bits) relative to the window's
//    lower end, clamped to 2048 bins.  A thousand squared errors spread over ~10 binades leave a handful of keys in the
//    selected bin; they are gathered and ranked by one wave.  sh.hist must be zero on entry and is left zero (the scan
//    phase clears the bins it reads), so no zeroing pass and five barriers in all;
//  - general path (selected bin clamped or holding more than 64 keys): MSB radix select with 11-bit digits, switching
//    to the same finisher as soon as the selected digit holds at most 64 keys.
// The 8-bit radix version needed 24 barriers per Gauss-Newton iteration and was ~45 % of the pose solve.
// (round 2c: 64 bins per binade over 2^-22 .. 2^10 instead of 16 over 2^-40 .. 2^88 — squared pixel errors live in that window,
//  and the finisher ranks its keys one broadcast at a time: ~100 cycles per key of the selected bin, 3 k cycles per call with the
//  ~30 keys a 16-per-binade bin holds around the median of a thousand)
#ifndef GS_KEY_MBITS
#define GS_KEY_MBITS 6
#endif
#define GS_KEY_LOW_EXP (GS_KEY_MBITS == 6 ? 22 : (GS_KEY_MBITS == 5 ? 40 : 40))   // window starts at 2^-LOW_EXP
#define GS_KEY_BASE ((1023 - GS_KEY_LOW_EXP) << GS_KEY_MBITS)
__device__ __forceinline__ int small_key_bin(unsigned long long key) {
    const int t = (int)(key >> (52 - GS_KEY_MBITS)) - GS_KEY_BASE;
    return t < 0 ? 0 : (t > GS_BINS - 1 ? GS_BINS - 1 : t);
}
// rank the keys for which `mine` holds among themselves (cnt <= 64 of them): returns the k-th smallest
template <int MPT, int THREADS>
__device__ double small_select_finish(GnSmallShared<THREADS, MPT>& sh, const unsigned long long key[MPT], const bool mine[MPT], int cnt, int k) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int q = 0; q < MPT; q++)
        if (mine[q]) sh.cand[atomicAdd(&sh.n_cand, 1)] = key[q];
    __syncthreads();
    if (wid == 0) {
        const unsigned long long me = lane < cnt ? sh.cand[lane] : ~0ull;
        int rank = 0;
        for (int j = 0; j < cnt; j++) {
            const unsigned long long o = sh.cand[j];
            rank += (o < me || (o == me && j < lane)) ? 1 : 0;
        }
        if (lane < cnt && rank == k) sh.cand[63] = me;   // exactly one lane (ties broken by index); read below
    }
    __syncthreads();
    return __longlong_as_double((long long)sh.cand[63]);
}
// block-wide scan of sh.hist (GS_BINS / GS_THREADS bins per thread); the owner of rank k publishes bin / residual rank /
// count and resets the candidate counter.  clear: zero the bins while reading them.
template <int MPT, int THREADS>
__device__ void small_select_scan(GnSmallShared<THREADS, MPT>& sh, int k, bool clear) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int BPT = GS_BINS / THREADS, GS_WAVES = THREADS / 64;
    unsigned c[BPT];
    int s = 0;
#pragma unroll
    for (int b = 0; b < BPT; b++) {
        c[b] = sh.hist[BPT * tid + b];
        if (clear) sh.hist[BPT * tid + b] = 0;
        s += (int)c[b];
    }
    const int incl = wave_incl_scan_i32(s);
    if (lane == 63) sh.scan[wid] = incl;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int w = 0; w < GS_WAVES; w++)
        if (w < wid) off += sh.scan[w];
    const int excl = off + incl - s;
    if (excl <= k && k < excl + s) {
        int kk = k - excl, dg = BPT * tid;
#pragma unroll
        for (int b = 0; b < BPT - 1; b++)
            if (kk >= (int)c[b] && dg == BPT * tid + b) {
                kk -= (int)c[b];
                dg++;
            }
        sh.sel_digit = dg;
        sh.sel_k = kk;
        sh.sel_cnt = (int)c[dg - BPT * tid];
        sh.n_cand = 0;
    }
    __syncthreads();
}

#ifdef K7_TIMING
static __device__ long long g_sel_ph[4];   // select sub-phases (timing build): histogram | scan | candidates + rank | -
#endif
#if defined(K7_TIMING) && defined(SEL_TIMING)   // (their stamps are read-modify-writes of global memory by thread 0: ~2 k cycles per
                                                //  select call of their own — only with -DSEL_TIMING, or the `select` phase is mostly them)
#define SEL_PH(i) { if (threadIdx.x == 0) { const long long n_ = (long long)__builtin_readcyclecounter(); g_sel_ph[i] += n_ - spt_; spt_ = n_; } }
#define SEL_PH0 long long spt_ = (long long)__builtin_readcyclecounter();
#else
#define SEL_PH(i)
#define SEL_PH0
#endif
// ---- round 4: the order statistic on the pose loops' critical path (it was a quarter of both loops) ------------------------
// wave maximum of a non-negative int, in every lane
__device__ __forceinline__ int wave_max_nonneg_i32(int v) {
    v = max(v, dpp_row_shr0_i32<1>(v));
    v = max(v, dpp_row_shr0_i32<2>(v));
    v = max(v, dpp_row_shr0_i32<4>(v));
    v = max(v, dpp_row_shr0_i32<8>(v));
    v = max(v, dpp_bcast_i32<0x142, 0xA>(0, v));
    v = max(v, dpp_bcast_i32<0x143, 0xC>(0, v));
    return __builtin_amdgcn_readlane(v, 63);
}
// The k-th smallest of the keys the first `cnt` lanes of a wave hold (one each, cnt <= 64, k < number of finite keys), by
// ranking in registers: every lane counts the keys below its own — key j travels through two v_readlane, no LDS, no barrier —
// and the answer is the key of greatest rank <= k (equal keys share a rank: exactly the value sorted[k] has).
template <int CNT_MAX>
__device__ __forceinline__ unsigned long long wave_rank_select(unsigned long long me, int cnt, int k) {
    const unsigned lo = (unsigned)me, hi = (unsigned)(me >> 32);
    int rank = 0;
    if (CNT_MAX == 64) {
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, j);
            rank += o < me ? 1 : 0;
        }
    } else {
        for (int j = 0; j < cnt; j++) {   // (cnt is wave-uniform)
            const unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, j);
            rank += o < me ? 1 : 0;
        }
    }
    const bool in = (int)(threadIdx.x & 63) < cnt && rank <= k;
    const int m = wave_max_nonneg_i32(in ? rank + 1 : 0) - 1;
    const unsigned long long pick = __builtin_amdgcn_ballot_w64(in && rank == m);
    const int src = __builtin_ctzll(pick);
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, src) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, src);
}
// bin b of the transposed histogram: lane L = b / 32 owns bins 32 L .. 32 L + 31, its j-th at word j * 64 + L (conflict-free for
// the lane-parallel scan below)
__device__ __forceinline__ int hist2_word(int b) { return (b & 31) * 64 + (b >> 5); }
// Two-barrier exact select for the 256-thread loops (THREADS * MPT keys in registers): LDS-atomic histogram | barrier | EVERY wave
// scans the histogram for itself (lane = 32 bins; the crossing lane's bins by a second 32-lane scan) — no meeting, no second
// barrier for the scan | the few keys of the selected bin are appended to a list | barrier | every wave ranks them in registers.
// `par` alternates between calls: a call clears the buffer of the call before.  Returns false (nothing consumed but the buffers,
// which it leaves clean) when the selected bin is a clamped end bin or holds more than 64 keys: the caller takes the general path.
// the histogram half of small_select_fast, for a caller that has a barrier of its own coming (the pose loop's error phase)
template <int MPT, int THREADS>
__device__ __forceinline__ void small_select_fast_hist(GnSmallShared<THREADS, MPT>& sh, const unsigned long long key[MPT], int n, int par) {
    const int tid = threadIdx.x;
    unsigned* h = sh.hist2[par];
#pragma unroll
    for (int q = 0; q < MPT; q++)
        if (tid + q * THREADS < n) atomicAdd(&h[hist2_word(small_key_bin(key[q]))], 1u);
    // (the other buffer: dirty from the call before, whose readers are long past their last barrier)
    for (int b = tid; b < GS_BINS; b += THREADS) sh.hist2[par ^ 1][b] = 0;
    if (tid == 0) sh.n_cand2[par] = 0;
}
template <int MPT, int THREADS, bool HIST_DONE = false>
__device__ __forceinline__ bool small_select_fast(GnSmallShared<THREADS, MPT>& sh, const unsigned long long key[MPT], int n, int k, int par, double* out) {
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned* h = sh.hist2[par];
    SEL_PH0
    int bin[MPT];
#pragma unroll
    for (int q = 0; q < MPT; q++) bin[q] = small_key_bin(key[q]);
    if (!HIST_DONE) {
        small_select_fast_hist<MPT, THREADS>(sh, key, n, par);
        __syncthreads();
    }
    SEL_PH(0)
    unsigned c[32];
    int ssum = 0;
    {   // (four partial sums: a chain of 32 dependent adds is 300 cycles of a phase every wave waits for)
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int j = 0; j < 32; j += 4) {
            c[j] = h[j * 64 + lane];
            c[j + 1] = h[(j + 1) * 64 + lane];
            c[j + 2] = h[(j + 2) * 64 + lane];
            c[j + 3] = h[(j + 3) * 64 + lane];
            s0 += (int)c[j];
            s1 += (int)c[j + 1];
            s2 += (int)c[j + 2];
            s3 += (int)c[j + 3];
        }
        ssum = (s0 + s1) + (s2 + s3);
    }
    const int incl = wave_incl_scan_i32(ssum), excl = incl - ssum;
    const unsigned long long own = __builtin_amdgcn_ballot_w64(excl <= k && k < incl);   // exactly one lane (k < total)
    const int L = __builtin_ctzll(own);
    const int k1 = k - __builtin_amdgcn_readlane(excl, L);
    // the crossing lane's 32 bins, one per lane of the lower half
    const int cj = lane < 32 ? (int)h[lane * 64 + L] : 0;
    const int incl2 = wave_incl_scan_i32(cj), excl2 = incl2 - cj;
    const unsigned long long own2 = __builtin_amdgcn_ballot_w64(lane < 32 && excl2 <= k1 && k1 < incl2);
    const int J = __builtin_ctzll(own2);
    const int sel = 32 * L + J, k2 = k1 - __builtin_amdgcn_readlane(excl2, J), cnt = __builtin_amdgcn_readlane(cj, J);
    const bool ok = sel != 0 && sel != GS_BINS - 1 && cnt <= 64;
    SEL_PH(1)
    if (ok) {
#pragma unroll
        for (int q = 0; q < MPT; q++)
            if (tid + q * THREADS < n && bin[q] == sel) sh.cand2[par][atomicAdd(&sh.n_cand2[par], 1)] = key[q];
    }
    __syncthreads();
    SEL_PH(2)
    if (!ok) return false;
    const unsigned long long me = lane < cnt ? sh.cand2[par][lane] : ~0ull;
    *out = __longlong_as_double((long long)wave_rank_select<0>(me, cnt, k2));
    SEL_PH(3)
    return true;
}
// MPT == 1 and n <= 128 (the coarse set): no histogram — every key is ranked against the others by broadcast reads
template <int MPT, int THREADS, bool HIST_DONE = false>
__device__ double small_select_kth(GnSmallShared<THREADS, MPT>& sh, int n, int k, int par) {
    if (MPT == 1 && THREADS == 64) {
        // one wave (the coarse set): every lane's key stays in its register (the slot in LDS is read back once)
        const unsigned long long me = (unsigned long long)__double_as_longlong(sh.keys[threadIdx.x]);
        return __longlong_as_double((long long)wave_rank_select<64>(me, 64, k));
    }
    if (MPT == 1 && n <= 128) {
        // One key per thread slot (slots past n hold +inf).  With n <= 64 (<= 128) only the first wave (two) holds
        // keys, so the list is cut in four (two) parts and thread (part, i) ranks key i against its part; the partial
        // ranks meet in LDS.  Sixteen independent broadcast reads per round; "less" and "equal" are counted separately —
        // the index tie-break costs as much as the comparison, ties are rare, and only a key that has an equal walks the
        // list again.
        const int tid = threadIdx.x;
        // (one wave, THREADS == 64: every lane ranks its key against the whole list, nothing to meet)
        const int parts = THREADS == 64 ? 1 : (n <= 64 ? 4 : 2), per = THREADS / parts;
        const int i = tid % per, part = tid / per;
        const int len = ((n + parts - 1) / parts + 15) & ~15, j_begin = part * len, j_end = min(j_begin + len, THREADS);
        const unsigned long long me = (unsigned long long)__double_as_longlong(sh.keys[i]);
        int rank = 0, n_eq = 0;
        for (int j0 = j_begin; j0 < j_end; j0 += 16) {
            unsigned long long o[16];
#pragma unroll
            for (int u = 0; u < 16; u++) o[u] = (unsigned long long)__double_as_longlong(sh.keys[j0 + u]);
#pragma unroll
            for (int u = 0; u < 16; u++) {
                rank += o[u] < me ? 1 : 0;
                n_eq += o[u] == me ? 1 : 0;
            }
        }
        if (parts > 1) {   // (the histogram is unused on this path — n is fixed for the launch — and serves as the meeting place)
            if (tid < per) sh.hist[tid] = 0, sh.hist[THREADS + tid] = 0;
            __syncthreads();
            atomicAdd(&sh.hist[i], (unsigned)rank);
            atomicAdd(&sh.hist[THREADS + i], (unsigned)n_eq);
            __syncthreads();
            rank = (int)sh.hist[i];
            n_eq = (int)sh.hist[THREADS + i];
        }
        if (tid < n && n_eq > 1)
            for (int j = 0; j < tid; j++) rank += (unsigned long long)__double_as_longlong(sh.keys[j]) == me ? 1 : 0;
        if (tid < n && rank == k) sh.cand[63] = me;   // exactly one thread (ties broken by index)
        __syncthreads();
        const double r = __longlong_as_double((long long)sh.cand[63]);
        __syncthreads();   // (read before the next iteration's keys / the next winner overwrite it)
        return r;
    }
    const int tid = threadIdx.x;
    unsigned long long key[MPT];
    bool mine[MPT];
    SEL_PH0
#pragma unroll
    for (int q = 0; q < MPT; q++) {
        const int i = tid + q * THREADS;
        key[q] = i < n ? (unsigned long long)__double_as_longlong(sh.keys[i]) : ~0ull;
    }
    {
        double r_;
        if (small_select_fast<MPT, THREADS, HIST_DONE>(sh, key, n, k, par, &r_)) return r_;
    }
#pragma unroll
    for (int q = 0; q < MPT; q++)
        if (tid + q * THREADS < n) atomicAdd(&sh.hist[small_key_bin(key[q])], 1u);
    __syncthreads();
    SEL_PH(0)
    small_select_scan<MPT, THREADS>(sh, k, true);
    SEL_PH(1)
    {
        const int bin = sh.sel_digit, cnt = sh.sel_cnt;
        if (bin != 0 && bin != GS_BINS - 1 && cnt <= 64) {
#pragma unroll
            for (int q = 0; q < MPT; q++) mine[q] = tid + q * THREADS < n && small_key_bin(key[q]) == bin;
            const double r_ = small_select_finish<MPT, THREADS>(sh, key, mine, cnt, sh.sel_k);
            SEL_PH(2)
            return r_;
        }
    }
    // general path
    unsigned long long prefix = 0;
    int top = 64;   // bits [top, 64) of the answer are fixed in `prefix`
    while (top > 0) {
        const int bits = top >= 11 ? 11 : top, shift = top - bits;
        const unsigned mask = (1u << bits) - 1u;
        __syncthreads();   // (the previous round's reads of sel_* are done; hist is zero)
#pragma unroll
        for (int q = 0; q < MPT; q++)
            if (tid + q * THREADS < n && (top == 64 || (key[q] >> top) == (prefix >> top)))
                atomicAdd(&sh.hist[(unsigned)(key[q] >> shift) & mask], 1u);
        __syncthreads();
        small_select_scan<MPT, THREADS>(sh, k, true);
        prefix |= (unsigned long long)sh.sel_digit << shift;
        k = sh.sel_k;
        top = shift;
        const int cnt = sh.sel_cnt;
        if (top > 0 && cnt <= 64) {
#pragma unroll
            for (int q = 0; q < MPT; q++) mine[q] = tid + q * THREADS < n && (key[q] >> top) == (prefix >> top);
            return small_select_finish<MPT, THREADS>(sh, key, mine, cnt, k);
        }
    }
    return __longlong_as_double((long long)prefix);
}

// CalcJacobian (include/Tracker.h:125-136) from the cached camera-frame point and derivatives.  The
// fast path does not store J: v3Cam / m2CamDerivs only change on non-linear iterations, so
// re-deriving J from them in the linear iterations gives the very values the reference keeps.
// (iz = 1.0 / Z is cached with the point: v3Cam only changes on non-linear iterations, the quotient is the same value)
__device__ __forceinline__ void small_jacobian(const double cam3[3], double iz, const double D[4], double J[12]) {
    const double X = cam3[0], Y = cam3[1], Z = cam3[2];
    const double gx[6] = {1, 0, 0, 0, Z, -Y};
    const double gy[6] = {0, 1, 0, -Z, 0, X};
    const double gz[6] = {0, 0, 1, Y, -X, 0};
#pragma unroll
    for (int m = 0; m < 6; m++) {
        const double mx = (gx[m] - X * gz[m] * iz) * iz;
        const double my = (gy[m] - Y * gz[m] * iz) * iz;
        J[m] = D[0] * mx + D[1] * my;
        J[6 + m] = D[2] * mx + D[3] * my;
    }
}

struct SmallMeas {
    double world[3], fnd[2], sn;
    double cam3[3], iz, img[2], D[4];
    double J[12];   // m26Jacobian of the last CalcJacobian (include/Tracker.h:125-136): refreshed when cam3 / D change, i.e. on
                    // nonlinear iterations only — with one wave per SIMD the 512-register file holds it for all four measurements
    int found;
    int listed;     // a measurement sits in this slot (found can still be cleared: not in the potentially-visible set)
};

// Where the loop's measurements come from.  The default: the arrays of ptam_pose_gn* (meas[i], entry[i], i < n).
struct PoseArrayLoader {
    const ptam_pose_meas* __restrict__ meas;
    const ptam_projection* __restrict__ entry;
    const int* __restrict__ n_dev;
    // false: nothing to do for this launch (the loader has said why to whoever needs to know)
    template <class SH, class T>
    __device__ __forceinline__ bool begin(SH&, int& n, int cap, T&) {
        if (n_dev) n = min(n, max(*n_dev, 0));   // counted variant: the measurement list was compacted on the device
        (void)cap;
        return true;
    }
    // measurement i into slot q of the thread: world, found position, noise scale and (has_entry) the TrackerData state
    __device__ __forceinline__ void load(int q, int i, int n, SmallMeas& t) const {
        (void)q;
        if (i < n) {
#pragma unroll
            for (int k = 0; k < 3; k++) t.world[k] = meas[i].world[k];
            t.fnd[0] = meas[i].found[0];
            t.fnd[1] = meas[i].found[1];
            t.sn = meas[i].sqrt_inv_noise;
            t.listed = 1;
            if (entry) {
#pragma unroll
                for (int k = 0; k < 3; k++) t.cam3[k] = entry[i].cam[k];
                t.img[0] = entry[i].image[0];
                t.img[1] = entry[i].image[1];
#pragma unroll
                for (int k = 0; k < 4; k++) t.D[k] = entry[i].derivs[k];
            }
        }
    }
    __device__ __forceinline__ bool has_entry() const { return entry != nullptr; }
    // where measurement i's TrackerData state goes back to (resident chain), given the chain's io block
    __device__ __forceinline__ ptam_projection* td_target(int q, int i, const PoseChainIo& io) const {
        (void)q;
        return (ptam_projection*)((char*)io.td_base + (size_t)(io.td_index ? io.td_index[i] : i) * io.td_stride);
    }
    __device__ __forceinline__ void td_also(int, int, const SmallMeas&) const {}
    template <class SH>
    __device__ __forceinline__ void publish(SH&) const {}
    __device__ __forceinline__ int listed_total(int n) const { return n; }
};

// atan for x >= 0 by argument reduction (c in {0, 1/2, 1, 3/2}; x >= 39/16: pi/2 - atan(1/x)) and the 11-term odd minimax
// polynomial of fdlibm on |t| < 7/16, <= 2 ulp against libm (the bundle kernels' form, ba_math.inc): ~45 VALU instructions
// where the library call is ~80 — on the one-wave coarse loop every instruction of the re-projection is on the frame's critical path
static __constant__ double POSE_ATAN_K[20] = {
    3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,  -1.11111104054623557880e-01,
    9.09088713343650656196e-02,  -7.69187620504482999495e-02, 6.66107313738753120669e-02,  -5.83357013379057348645e-02,
    4.97687799461593236017e-02,  -3.65315727442169155270e-02, 1.62858201153657823623e-02,
    4.63647609000806093515e-01,  7.85398163397448278999e-01,  9.82793723247329054082e-01,  1.57079632679489655800e+00,
    0.4375, 0.6875, 1.1875, 2.4375, 1.5};
__device__ __forceinline__ double pose_atan_pos(double x) {
    const double* __restrict__ K = POSE_ATAN_K;
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
    double p1 = fma(w, K[10], K[8]);
    p1 = fma(w, p1, K[6]);
    p1 = fma(w, p1, K[4]);
    p1 = fma(w, p1, K[2]);
    p1 = fma(w, p1, K[0]);
    double p2 = fma(w, K[9], K[7]);
    p2 = fma(w, p2, K[5]);
    p2 = fma(w, p2, K[3]);
    p2 = fma(w, p2, K[1]);
    return hi - (t * (z * p1 + w * p2) - t);
}

// TrackerData::Project with the pose in LDS; updates the cached state exactly like td_project.  Same formulas as cam_project /
// cam_derivs (common.h) with the seven IEEE divisions of a projection — a dependent chain of a dozen instructions each — as
// Newton-refined reciprocals (1 / Z, 1 / r, 1 / (r^2 (1 + k^2 r^2)): <= 1 ulp each) and the atan above.
// Returns "camera model reached" (include/Tracker.h:70-85 bails out before it for a point behind the camera or outside the model's
// radius; ProjectAndDerivs :89-94 then reads the camera's cache of ANOTHER point's projection — here the point keeps its own
// derivatives, and the event is counted: g_pose_hazards, ptam_ctx_cache_hazards).
static __device__ unsigned long long g_pose_hazards;
__device__ __forceinline__ bool small_project(const DevCam& cam, const double* pose, SmallMeas& t, bool& in_image) {
    in_image = false;
    se3_apply(pose, t.world[0], t.world[1], t.world[2], t.cam3[0], t.cam3[1], t.cam3[2]);
    t.iz = rcp_nr(t.cam3[2]);
    if (t.cam3[2] < 0.001) return false;
    const double x = t.cam3[0] * t.iz, y = t.cam3[1] * t.iz;
    const double r2 = x * x + y * y;
    if (r2 > cam.largest_radius * cam.largest_radius) return false;
    const double r = sqrt(r2);
    const bool small_r = r < 0.001 || cam.w == 0.0;
    const double ir = rcp_nr(small_r ? 1.0 : r);
    const double f = small_r ? 1.0 : cam.w_inv * pose_atan_pos(r * cam.two_tan) * ir;
    const double u = cam.cx + cam.fx * (f * x), v = cam.cy + cam.fy * (f * y);
    t.img[0] = u;
    t.img[1] = v;
    {
        // GetProjectionDerivs src/ATANCamera.cc:179-209
        const double k = cam.two_tan, rd = r * cam.dist_enabled;
        double dx = 0.0, dy = 0.0;
        if (!(rd < 0.01)) {
            const double rr = rd * rd;
            const double inv_den = rcp_nr(rr * (1 + k * k * rr)), irr = rcp_nr(rr);
            dx = cam.w_inv * (k * x) * inv_den - x * f * irr;
            dy = cam.w_inv * (k * y) * inv_den - y * f * irr;
        }
        t.D[0] = cam.fx * (dx * x + f);
        t.D[2] = cam.fy * (dx * y);
        t.D[1] = cam.fx * (dy * x);
        t.D[3] = cam.fy * (dy * y + f);
    }
    if (r > cam.max_r) return true;
    if (u < 0 || v < 0 || u > cam.width || v > cam.height) return true;
    in_image = true;
    return true;
}

// Fast path, n <= 1024: 256 threads x 4 measurements held in registers (one wave per SIMD, the four
// independent measurements of a thread give the fp64 pipeline its ILP), e^2 keys of the order
// statistic in LDS, wave sums by DPP, same arithmetic and reduction order as the general kernel.
template <int MPT, int THREADS, class LOADER>
__device__ __forceinline__ void pose_gn_small_body(const DevCam& cam, int n, LOADER& ld,
                                                   double* __restrict__ pose_io, const ptam_gn_opts& opts,
                                                   int* __restrict__ flags, double* __restrict__ updates,
                                                   ulonglong2* __restrict__ host_slots, unsigned long long seq,
                                                   const int* __restrict__ n_dev, const PoseIn* pin, const PoseChainIo& io, int size_guard) {
    typedef GnSmallShared<THREADS, MPT> Sh;
    constexpr int GS_WAVES = Sh::WAVES, GS_SLICES = Sh::SLICES;
    __shared__ Sh sh;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (size_guard == 1 && *n_dev > THREADS * MPT) return;   // (the general kernel, enqueued behind this one, takes the long list)
    if (size_guard == 3 && *n_dev > THREADS * MPT) {         // (nobody is behind me: tell the host to send the general kernel)
        if (threadIdx.x == 0 && io.result_seq) *(volatile unsigned long long*)io.result_seq = io.seq | POSE_CHAIN_LONG;
        return;
    }
#ifdef K7_TIMING
    const long long tk0_ = (long long)__builtin_readcyclecounter();
#endif
    SmallMeas t[MPT];
#pragma unroll
    for (int q = 0; q < MPT; q++) {
        t[q].found = 0;
        t[q].listed = 0;
        t[q].cam3[0] = t[q].cam3[1] = 0;
        t[q].cam3[2] = 1;
        t[q].iz = 1;
        t[q].img[0] = t[q].img[1] = 0;
        t[q].D[0] = t[q].D[1] = t[q].D[2] = t[q].D[3] = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) t[q].J[k] = 0;
        t[q].world[0] = t[q].world[1] = t[q].world[2] = t[q].fnd[0] = t[q].fnd[1] = t[q].sn = 0;
    }
    if (tid < 12) sh.pose[tid] = (pin && pin->use) ? pin->v[tid] : pose_io[tid];   // (pin: a kernel argument — a local copy indexed by tid would live in scratch memory)
    if (tid < 6) sh.mu[tid] = 0;
    if (!ld.begin(sh, n, THREADS * MPT, t)) return;
#ifdef K7_TIMING
    const long long tkA_ = (long long)__builtin_readcyclecounter();
#endif
    if (THREADS > 64)   // (the one-wave kernel ranks in registers: no histogram.  Its zeroing loop was 2.4 k cycles of the coarse stage)
        for (int b = tid; b < GS_BINS; b += THREADS) sh.hist[b] = sh.hist2[0][b] = sh.hist2[1][b] = 0;   // the selects keep them zero between calls
#pragma unroll
    for (int q = 0; q < MPT; q++) ld.load(q, tid + q * THREADS, n, t[q]);
    __syncthreads();
#ifdef K7_TIMING
    const long long tkB_ = (long long)__builtin_readcyclecounter();
#endif
#pragma unroll
    for (int q = 0; q < MPT; q++) {
        const int i = tid + q * THREADS;
        if (i < n && flags) flags[i] = 0;
        if (t[q].listed) {
            t[q].found = 1;
            if (ld.has_entry()) {
                t[q].iz = rcp_nr(t[q].cam3[2]);
            } else {
                bool in_image;
                small_project(cam, sh.pose, t[q], in_image);
                if (!in_image) t[q].found = 0;   // not in the potentially-visible set (src/Tracker.cc:456-458)
            }
            if (t[q].found) small_jacobian(t[q].cam3, t[q].iz, t[q].D, t[q].J);   // (stays zero otherwise)
        }
    }
#ifdef K7_TIMING
    const long long tk1_ = (long long)__builtin_readcyclecounter();
    long long ph[6] = {0, 0, 0, 0, 0, 0};
#define PH(i) { const long long n_ = (long long)__builtin_readcyclecounter(); ph[i] += n_ - pt_; pt_ = n_; }
#else
#define PH(i)
#endif
    int sel_par = 0;
    int hazards = 0;   // re-projections of found measurements that bailed out before the camera model (see small_project)
    for (int iter = 0; iter < opts.iterations; iter++) {
#ifdef K7_TIMING
        long long pt_ = (long long)__builtin_readcyclecounter();
#endif
        const bool nonlinear = (opts.nonlinear_mask >> iter) & 1u;
        const double ov = iter > opts.override_after ? opts.override_sigma_sq : 0.0;
        double ex[MPT], ey[MPT], e2[MPT];
        int cnt = 0;
        // The four measurements of a thread are kept in straight-line code (selects instead of per-measurement branches):
        // with one wave per SIMD the only latency hiding there is comes from interleaving their dependence chains, and
        // an exec-mask branch per measurement fences the scheduler.  Slots without a measurement hold benign values
        // (zero Jacobian, zero noise scale) and are masked out of the keys, the count and the weights.
        if (iter != 0 && nonlinear) {
#pragma unroll
            for (int q = 0; q < MPT; q++)
                if (t[q].found) {
                    bool in_image;
                    if (!small_project(cam, sh.pose, t[q], in_image)) hazards++;
                    small_jacobian(t[q].cam3, t[q].iz, t[q].D, t[q].J);
                }
        } else if (iter != 0) {   // LinearUpdate include/Tracker.h:139-142
            double mu[6];
#pragma unroll
            for (int m = 0; m < 6; m++) mu[m] = sh.mu[m];
#pragma unroll
            for (int q = 0; q < MPT; q++) {
                double a = 0, b = 0;
#pragma unroll
                for (int m = 0; m < 6; m++) {
                    a += t[q].J[m] * mu[m];
                    b += t[q].J[6 + m] * mu[m];
                }
                t[q].img[0] += a;   // (J == 0 where there is no measurement)
                t[q].img[1] += b;
            }
        }
#pragma unroll
        for (int q = 0; q < MPT; q++) {
            // CalcPoseUpdate :946-954
            ex[q] = t[q].sn * (t[q].fnd[0] - t[q].img[0]);
            ey[q] = t[q].sn * (t[q].fnd[1] - t[q].img[1]);
            e2[q] = ex[q] * ex[q] + ey[q] * ey[q];
            cnt += t[q].found;
            if (!(ov > 0)) sh.keys[tid + q * THREADS] = t[q].found ? e2[q] : __longlong_as_double(0x7ff0000000000000ll);
        }
        if (MPT > 1 && !(ov > 0)) {
            // the select's histogram rides on this phase's barrier (one barrier and an LDS round trip of the keys less per call)
            unsigned long long kk[MPT];
#pragma unroll
            for (int q = 0; q < MPT; q++) kk[q] = t[q].found ? (unsigned long long)__double_as_longlong(e2[q]) : 0x7ff0000000000000ull;
            small_select_fast_hist<MPT, THREADS>(sh, kk, n, sel_par);
        }
        cnt = wave_sum_i32(cnt);
        if (lane == 0) sh.wcount[wid] = cnt;
        __syncthreads();   // also: every thread is done reading sh.mu (linear update) and sh.pose
        PH(0)
        int nf = 0;
#pragma unroll
        for (int i = 0; i < GS_WAVES; i++) nf += sh.wcount[i];
        if (nf > 0) {
            double sigma_sq;
            if (ov > 0)
                sigma_sq = ov;
            else {
                const double med = small_select_kth<MPT, THREADS, (MPT > 1)>(sh, n, nf / 2, sel_par);
                sel_par ^= 1;
                // FindSigmaSquared (include/Tools.h:128-162) with its division as a Newton-refined reciprocal: the formula sits on the
                // iteration's critical path in every thread, and an IEEE fp64 division is a dependent chain of a dozen instructions
                const unsigned long long den = (unsigned long long)nf * 2ull - 6ull;   // wraps for n < 3 like the reference
                double sigma = 1.4826 * (1 + 5.0 * rcp_nr((double)den)) * sqrt(med);
                sigma = (opts.estimator == PTAM_EST_HUBER ? 1.345 : 4.6851) * sigma;
                sigma_sq = sigma * sigma;
            }
            PH(1)
            const double inv_sigma_sq = rcp_nr(sigma_sq);   // (one reciprocal for the workgroup's weights instead of a division per measurement)
            // WLS<6> :973-1002: C += (w J_r)(J_r)^T, b += e_r (w J_r), J_r scaled by dSqrtInvNoise
            double acc[27];
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] = 0;
#pragma unroll
            for (int q = 0; q < MPT; q++) {
                // weight 0 (an outlier, or no measurement in this slot): every product below is an exact zero
                double wgt;   // Weight() of include/Tools.h:128-228 with e^2 / sigma^2 as a product
                if (opts.estimator == PTAM_EST_TUKEY) {
                    const double r = e2[q] > sigma_sq ? 0.0 : 1.0 - e2[q] * inv_sigma_sq;
                    wgt = r * r;
                } else if (opts.estimator == PTAM_EST_CAUCHY)
                    wgt = 1.0 / (1.0 + e2[q] * inv_sigma_sq);
                else
                    wgt = e2[q] < sigma_sq ? 1.0 : sqrt(sigma_sq / e2[q]);
                wgt = t[q].found ? wgt : 0.0;
                const double* Jm = t[q].J;
                const double er[2] = {ex[q], ey[q]};
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    double J[6], Jw[6];
#pragma unroll
                    for (int m = 0; m < 6; m++) {
                        J[m] = t[q].sn * Jm[r * 6 + m];
                        Jw[m] = J[m] * wgt;
                    }
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int b = 0; b <= a; b++) acc[k++] += Jw[a] * J[b];
#pragma unroll
                    for (int a = 0; a < 6; a++) acc[21 + a] += er[r] * Jw[a];
                }
                if (iter == opts.mark_outliers_iter && flags && t[q].found && wgt == 0.0) flags[tid + q * THREADS] = 1;
            }
            // 27 sums over 256 threads through LDS: every thread drops its partials column-wise, then 27 x 8 threads
            // each add a 32-thread slice and the 8 slices of a sum meet by shuffles — ~100 instructions per thread
            // instead of 27 six-step DPP reductions (~490) plus a serial four-wave combine (fixed order: deterministic)
            const int tcol = (tid >> 5) * 33 + (tid & 31);
#pragma unroll
            for (int k = 0; k < 27; k++) sh.tr[k][tcol] = acc[k];
        }
        PH(2)
        __syncthreads();
        PH(3)
        if (nf > 0 && tid < 27 * GS_SLICES) {
            const int k = tid / GS_SLICES, part = tid % GS_SLICES;
            const double* src = &sh.tr[k][part * 33];
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int j = 0; j < 32; j += 4) {
                a0 += src[j];
                a1 += src[j + 1];
                a2 += src[j + 2];
                a3 += src[j + 3];
            }
            double v = (a0 + a1) + (a2 + a3);
            v += dpp_row_shr_f64<1>(v);   // the slices of a sum (8, or 2 in the one-wave form) sit in consecutive lanes of a DPP row: the last collects
            if (GS_SLICES > 2) {
                v += dpp_row_shr_f64<2>(v);
                v += dpp_row_shr_f64<4>(v);
            }
            if (GS_SLICES > 8) v += dpp_row_shr_f64<8>(v);
            if (part == GS_SLICES - 1) sh.red[0][k] = v;
        }
        __syncthreads();
        PH(4)
        if (wid == 0) {
            // The 6 x 6 solve and exp() on EVERY lane of wave 0 (an instruction costs the wave the same whether one lane or all
            // of them are enabled), so that the last step — exp(x) * pose, 36 products, twelve LDS reads and twelve writes on
            // one thread — is three products, three reads and one write on lanes 0..11: lane 3 r + c forms R'[r][c], lane
            // 9 + r the translation's row r (the same sums in the same order as se3_exp_mul).
            double x[6] = {0, 0, 0, 0, 0, 0};
            if (nf > 0) {
                double C[36], b[6];
                int k = 0;
                for (int a = 0; a < 6; a++)
                    for (int c = 0; c <= a; c++) {
                        C[a * 6 + c] = C[c * 6 + a] = sh.red[0][k];
                        k++;
                    }
                for (int a = 0; a < 6; a++) {
                    C[a * 6 + a] += opts.prior;   // add_prior :974
                    b[a] = sh.red[0][21 + a];
                }
                ldlt6_solve(C, b, x);
            }
            double R[9], et[3];
            se3_exp_parts<true>(x, R, et);   // mse3CamFromWorld = SE3<>::exp(v6Update) * mse3CamFromWorld
            const int orow = lane < 9 ? lane / 3 : min(lane - 9, 2), ocol = lane < 9 ? lane - 3 * (lane / 3) : 0;
            const double r0 = orow == 0 ? R[0] : (orow == 1 ? R[3] : R[6]), r1 = orow == 0 ? R[1] : (orow == 1 ? R[4] : R[7]),
                         r2 = orow == 0 ? R[2] : (orow == 1 ? R[5] : R[8]);
            const double* Tc = lane < 9 ? sh.pose + ocol : sh.pose + 9;   // column ocol of the rotation (rows 3 apart), or the translation (rows 1 apart)
            const int ts = lane < 9 ? 3 : 1;
            const double t0 = Tc[0], t1 = Tc[ts], t2 = Tc[2 * ts];
            double o = r0 * t0 + r1 * t1 + r2 * t2;
            if (lane >= 9) o = (orow == 0 ? et[0] : (orow == 1 ? et[1] : et[2])) + o;
            __builtin_amdgcn_wave_barrier();   // (every lane has read the old pose)
            if (lane < 12) sh.pose[lane] = o;
            if (lane == 0) {
                for (int k = 0; k < 6; k++) sh.mu[k] = x[k];
                if (updates)
                    for (int k = 0; k < 6; k++) updates[6 * iter + k] = x[k];
            }
        }
        __syncthreads();
        PH(5)
    }
#ifdef K7_TIMING
    if (tid == 0 && io.result_seq == nullptr && io.td_base && (clock64() & 0xf000) == 0)
        printf("coarse pose kernel (%d threads): begin %lld | zero + barrier %lld | jacobian %lld | loop %lld (phases %lld %lld %lld %lld %lld %lld)\n", THREADS,
               tkA_ - tk0_, tkB_ - tkA_, tk1_ - tkB_, (long long)__builtin_readcyclecounter() - tk1_, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5]);
    if (tid == 0 && io.result_seq && (io.seq & 127) == 0)
        printf("fine pose kernel (%d threads x %d): begin %lld | zero + barrier %lld | jacobian %lld | loop %lld (phases %lld %lld %lld %lld %lld %lld)\n", THREADS, MPT,
               tkA_ - tk0_, tkB_ - tkA_, tk1_ - tkB_, (long long)__builtin_readcyclecounter() - tk1_, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5]);
    if (tid == 0 && io.result_seq && (io.seq & 127) == 0) {
        printf("   fine select sub-phases: histogram + barrier %lld | scans %lld | append + barrier %lld | rank %lld\n", g_sel_ph[0], g_sel_ph[1], g_sel_ph[2], g_sel_ph[3]);
    }
    if (tid == 0 && io.result_seq) g_sel_ph[0] = g_sel_ph[1] = g_sel_ph[2] = g_sel_ph[3] = 0;
    if (tid == 0 && updates)
        for (int i = 0; i < 6; i++) updates[6 * 20 + i] = (double)ph[i];
    if (tid == 0 && updates)
        for (int i = 0; i < 4; i++) {
            updates[6 * 20 + 6 + i] = (double)g_sel_ph[i];
            g_sel_ph[i] = 0;
        }
#endif
    if (tid < 12) pose_io[tid] = sh.pose[tid];
    if (hazards) atomicAdd(&g_pose_hazards, (unsigned long long)hazards);   // (rare: a tracked point that left the camera model)
    // resident chain: the measurements' TrackerData state goes back to the per-point table, scene depth sums
    if (io.td_base) {
#pragma unroll
        for (int q = 0; q < MPT; q++) {
            const int i = tid + q * THREADS;
            if (t[q].listed) {
                ptam_projection* o = ld.td_target(q, i, io);
#pragma unroll
                for (int k = 0; k < 3; k++) o->cam[k] = t[q].cam3[k];
                o->image[0] = t[q].img[0];
                o->image[1] = t[q].img[1];
#pragma unroll
                for (int k = 0; k < 4; k++) o->derivs[k] = t[q].D[k];
                ld.td_also(q, i, t[q]);
            }
        }
    }
    if (io.depth_out) {
        double z1 = 0, z2 = 0;
#pragma unroll
        for (int q = 0; q < MPT; q++) {
            const double z = t[q].listed ? t[q].cam3[2] : 0.0;
            z1 += z;
            z2 += z * z;
        }
        z1 = wave_sum_f64(z1);
        z2 = wave_sum_f64(z2);
        __syncthreads();
        if (lane == 0) {
            sh.red[wid][0] = z1;
            sh.red[wid][1] = z2;
        }
        __syncthreads();
        if (tid == 0) {
            double a = 0, b = 0;
            for (int w = 0; w < GS_WAVES; w++) {
                a += sh.red[w][0];
                b += sh.red[w][1];
            }
            io.depth_out[0] = a;
            io.depth_out[1] = b;
            io.depth_out[2] = (double)ld.listed_total(n);
            if (io.result_depth) {
                io.result_depth[0] = a;
                io.result_depth[1] = b;
                io.result_depth[2] = (double)ld.listed_total(n);
            }
        }
    }
    if (io.result_seq) {   // the frame's last kernel: pose, then the sequence word the host spins on (host-mapped memory)
        __syncthreads();
        ld.publish(sh);   // (what the loader booked at the kernel's start: every store to host memory happens here, once)
        if (threadIdx.x < 12) io.result_pose[threadIdx.x] = sh.pose[threadIdx.x];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *(volatile unsigned long long*)io.result_seq = io.seq;
    }
    // the refined pose also goes straight into host-mapped memory as (word, sequence) pairs the host spins on: the call
    // returns one PCIe write after the last iteration instead of a D2H copy plus a stream synchronisation later
    if (host_slots && tid < 12) host_slots[tid] = make_ulonglong2((unsigned long long)__double_as_longlong(sh.pose[tid]), seq);
}

