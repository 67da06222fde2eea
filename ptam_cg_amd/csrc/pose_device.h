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
static __device__ void ldlt6_solve(double A[36], const double b[6], double x[6]) {
    double inv_d[6];
#pragma unroll
    for (int col = 0; col < 6; col++) {
        double inv_diag = 1;
#pragma unroll
        for (int row = col; row < 6; row++) {
            double val = A[row * 6 + col];
#pragma unroll
            for (int c2 = 0; c2 < col; c2++) val -= A[c2 * 6 + col] * A[row * 6 + c2];
            if (row == col) {
                A[row * 6 + col] = val;
                inv_diag = rcp_nr(val);
                inv_d[col] = inv_diag;
            } else {
                A[col * 6 + row] = val;
                A[row * 6 + col] = val * inv_diag;
            }
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double val = b[i];
#pragma unroll
        for (int j = 0; j < i; j++) val -= A[i * 6 + j] * y[j];
        y[i] = val;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] *= inv_d[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double val = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; j++) val -= A[j * 6 + i] * x[j];
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
#ifndef GS_THREADS
#define GS_THREADS 256   // the workgroup of the 1024- and 256-measurement instantiations
#define GS_MPT 4         // measurements per thread of the largest one: n <= 1024
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

#ifdef K7_TIMING
static __device__ long long g_sel_ph[4];   // select sub-phases (timing build): histogram | scan | candidates + rank | -
#define SEL_PH(i) { if (threadIdx.x == 0) { const long long n_ = (long long)__builtin_readcyclecounter(); g_sel_ph[i] += n_ - spt_; spt_ = n_; } }
#define SEL_PH0 long long spt_ = (long long)__builtin_readcyclecounter();
#else
#define SEL_PH(i)
#define SEL_PH0
#endif
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
// MPT == 1 and n <= 128 (the coarse set): no histogram — every key is ranked against the others by broadcast reads
template <int MPT, int THREADS>
__device__ double small_select_kth(GnSmallShared<THREADS, MPT>& sh, int n, int k) {
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
        if (i < n) atomicAdd(&sh.hist[small_key_bin(key[q])], 1u);
    }
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
    __device__ __forceinline__ int listed_total(int n) const { return n; }
};

// TrackerData::Project with the pose in LDS; updates the cached state exactly like td_project
__device__ __forceinline__ void small_project(const DevCam& cam, const double* pose, SmallMeas& t, bool& in_image) {
    in_image = false;
    se3_apply(pose, t.world[0], t.world[1], t.world[2], t.cam3[0], t.cam3[1], t.cam3[2]);
    t.iz = 1.0 / t.cam3[2];
    if (t.cam3[2] < 0.001) return;
    const double x = t.cam3[0] / t.cam3[2], y = t.cam3[1] / t.cam3[2];
    if (x * x + y * y > cam.largest_radius * cam.largest_radius) return;
    double u, v, r, f;
    cam_project(cam, x, y, u, v, r, f);
    t.img[0] = u;
    t.img[1] = v;
    cam_derivs(cam, x, y, r, f, t.D);
    if (r > cam.max_r) return;
    if (u < 0 || v < 0 || u > cam.width || v > cam.height) return;
    in_image = true;
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
    for (int b = tid; b < GS_BINS; b += THREADS) sh.hist[b] = 0;   // small_select_kth keeps it zero between calls
#pragma unroll
    for (int q = 0; q < MPT; q++) ld.load(q, tid + q * THREADS, n, t[q]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MPT; q++) {
        const int i = tid + q * THREADS;
        if (i < n && flags) flags[i] = 0;
        if (t[q].listed) {
            t[q].found = 1;
            if (ld.has_entry()) {
                t[q].iz = 1.0 / t[q].cam3[2];
            } else {
                bool in_image;
                small_project(cam, sh.pose, t[q], in_image);
                if (!in_image) t[q].found = 0;   // not in the potentially-visible set (src/Tracker.cc:456-458)
            }
            if (t[q].found) small_jacobian(t[q].cam3, t[q].iz, t[q].D, t[q].J);   // (stays zero otherwise)
        }
    }
#ifdef K7_TIMING
    long long ph[6] = {0, 0, 0, 0, 0, 0};
#define PH(i) { const long long n_ = (long long)__builtin_readcyclecounter(); ph[i] += n_ - pt_; pt_ = n_; }
#else
#define PH(i)
#endif
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
                    small_project(cam, sh.pose, t[q], in_image);
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
                const double med = small_select_kth<MPT, THREADS>(sh, n, nf / 2);
                sigma_sq = est_sigma_sq_from_median(opts.estimator, med, (unsigned long long)nf);
            }
            PH(1)
            const double inv_sigma_sq = 1.0 / sigma_sq;   // (one division for the workgroup's weights instead of one per measurement)
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
        if (tid == 0) {
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
            double np[12];
            se3_exp_mul<true>(x, sh.pose, np);   // mse3CamFromWorld = SE3<>::exp(v6Update) * mse3CamFromWorld
            for (int k = 0; k < 12; k++) sh.pose[k] = np[k];
            for (int k = 0; k < 6; k++) sh.mu[k] = x[k];
            if (updates)
                for (int k = 0; k < 6; k++) updates[6 * iter + k] = x[k];
        }
        __syncthreads();
        PH(5)
    }
#ifdef K7_TIMING
    if (tid == 0 && updates)
        for (int i = 0; i < 6; i++) updates[6 * 20 + i] = (double)ph[i];
    if (tid == 0 && updates)
        for (int i = 0; i < 4; i++) {
            updates[6 * 20 + 6 + i] = (double)g_sel_ph[i];
            g_sel_ph[i] = 0;
        }
#endif
    if (tid < 12) pose_io[tid] = sh.pose[tid];
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
        if (threadIdx.x < 12) io.result_pose[threadIdx.x] = sh.pose[threadIdx.x];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *(volatile unsigned long long*)io.result_seq = io.seq;
    }
    // the refined pose also goes straight into host-mapped memory as (word, sequence) pairs the host spins on: the call
    // returns one PCIe write after the last iteration instead of a D2H copy plus a stream synchronisation later
    if (host_slots && tid < 12) host_slots[tid] = make_ulonglong2((unsigned long long)__double_as_longlong(sh.pose[tid]), seq);
}

