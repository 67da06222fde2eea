// pose.hip — Tracker pose Gauss-Newton on gfx950 (K4).
//   driver      src/Tracker.cc:613-643 (fine stage; :552-568 coarse stage via opts)
//   per point   include/Tracker.h:70-142 (Project / ProjectAndDerivs / CalcJacobian / LinearUpdate)
//   update      src/Tracker.cc:928-1005 (CalcPoseUpdate: Tukey sigma from the exact order statistic
//               sorted[n/2], weighted 6x6 normal equations with prior 100, LDL^T solve)
// The whole 10-iteration loop runs in ONE launch of ONE persistent workgroup (1024 threads): the
// pose lives in LDS, per-measurement state in an L2-resident scratch array, the median is an exact
// 8x8-bit MSB radix select over LDS histograms, the 27 normal-equation sums are reduced by wavefront
// shuffles + a fixed-order cross-wave pass (deterministic), thread 0 solves and applies exp().
#include <atomic>

#include "common.h"
#include "track_internal.h"
#include "pose_device.h"

struct PoseState {
    double cam[3];
    double img[2];
    double D[4];
    double J[12];
    double e[2];
    double e2;
    int found;
    int pad_;
};

#define GN_THREADS 1024
#define GN_WAVES (GN_THREADS / 64)

struct GnShared {
    double pose[12];
    double mu[6];
    double red[GN_WAVES][27];
    double sigma_sq;
    unsigned hist[256];
    int sel_digit;
    int sel_k;
    int count;
    int wcount[GN_WAVES];
};

// exact k-th smallest (0-based) of the e2 of found measurements: MSB-first radix select on the IEEE
// bit pattern (all keys >= 0, so unsigned bit order == value order)
__device__ double block_select_kth(GnShared& sh, const PoseState* __restrict__ st, int n, int k) {
    unsigned long long prefix = 0;
    const int tid = threadIdx.x;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) sh.hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += GN_THREADS) {
            if (!st[i].found) continue;
            const unsigned long long key = (unsigned long long)__double_as_longlong(st[i].e2);
            if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
                atomicAdd(&sh.hist[(key >> shift) & 255], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const unsigned c0 = sh.hist[4 * tid], c1 = sh.hist[4 * tid + 1], c2 = sh.hist[4 * tid + 2],
                           c3 = sh.hist[4 * tid + 3];
            const int s = (int)(c0 + c1 + c2 + c3);
            const int incl = wave_incl_scan_i32(s);
            const int excl = incl - s;
            if (excl <= k && k < incl) {
                int kk = k - excl, d = 4 * tid;
                if (kk >= (int)c0) {
                    kk -= c0;
                    d++;
                    if (kk >= (int)c1) {
                        kk -= c1;
                        d++;
                        if (kk >= (int)c2) {
                            kk -= c2;
                            d++;
                        }
                    }
                }
                sh.sel_digit = d;
                sh.sel_k = kk;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)sh.sel_digit << shift;
        k = sh.sel_k;
        __syncthreads();
    }
    return __longlong_as_double((long long)prefix);
}

// Tracker::CalcPoseUpdate (src/Tracker.cc:928-1005) for the whole block; result in sh.mu
__device__ void block_calc_pose_update(GnShared& sh, PoseState* __restrict__ st, int n,
                                       const double* __restrict__ found, const double* __restrict__ snoise,
                                       int found_stride, double override_sigma, int est, double prior, bool mark,
                                       int* __restrict__ flags) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // errors :946-954
    int cnt = 0;
    for (int i = tid; i < n; i += GN_THREADS) {
        if (!st[i].found) continue;
        const double s = snoise[(size_t)i * found_stride];
        const double ex = s * (found[(size_t)i * found_stride] - st[i].img[0]);
        const double ey = s * (found[(size_t)i * found_stride + 1] - st[i].img[1]);
        st[i].e[0] = ex;
        st[i].e[1] = ey;
        st[i].e2 = ex * ex + ey * ey;
        cnt++;
    }
    cnt = wave_sum_i32(cnt);
    if (lane == 0) sh.wcount[wid] = cnt;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int i = 0; i < GN_WAVES; i++) t += sh.wcount[i];
        sh.count = t;
    }
    __syncthreads();
    const int nf = sh.count;
    if (nf == 0) {   // :955-956
        if (tid < 6) sh.mu[tid] = 0;
        __syncthreads();
        return;
    }
    double sigma_sq;
    if (override_sigma > 0)
        sigma_sq = override_sigma;
    else {
        const double med = block_select_kth(sh, st, n, nf / 2);
        sigma_sq = est_sigma_sq_from_median(est, med, (unsigned long long)nf);
    }
    // WLS<6> accumulate :973-1002 : 21 lower-triangle sums of C + 6 of b
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; k++) acc[k] = 0;
    for (int i = tid; i < n; i += GN_THREADS) {
        if (!st[i].found) continue;
        const double wgt = est_weight(est, st[i].e2, sigma_sq);
        if (wgt == 0.0) {
            if (mark && flags) flags[i] = 1;
            continue;
        }
        const double s = snoise[(size_t)i * found_stride];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            double J[6], Jw[6];
#pragma unroll
            for (int m = 0; m < 6; m++) {
                J[m] = s * st[i].J[r * 6 + m];
                Jw[m] = J[m] * wgt;
            }
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b <= a; b++) acc[k++] += Jw[a] * J[b];
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] += st[i].e[r] * Jw[a];
        }
    }
#pragma unroll
    for (int k = 0; k < 27; k++) {
        const double v = wave_sum_f64(acc[k]);
        if (lane == 0) sh.red[wid][k] = v;
    }
    __syncthreads();
    if (tid < 27) {
        double t = 0;
        for (int i = 0; i < GN_WAVES; i++) t += sh.red[i][tid];
        sh.red[0][tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        double C[36], b[6], x[6];
        int k = 0;
        for (int a = 0; a < 6; a++)
            for (int c = 0; c <= a; c++) {
                C[a * 6 + c] = C[c * 6 + a] = sh.red[0][k++];
            }
        for (int a = 0; a < 6; a++) {
            C[a * 6 + a] += prior;   // add_prior :974
            b[a] = sh.red[0][21 + a];
        }
        ldlt6_solve(C, b, x);
        for (int a = 0; a < 6; a++) sh.mu[a] = x[a];
    }
    __syncthreads();
}

// TrackerData::Project (include/Tracker.h:70-85); returns "camera model reached"
__device__ __forceinline__ bool td_project(const DevCam& cam, const double* T, const double* X, PoseState& st,
                                           bool& in_image) {
    in_image = false;
    se3_apply(T, X[0], X[1], X[2], st.cam[0], st.cam[1], st.cam[2]);
    if (st.cam[2] < 0.001) return false;
    const double x = st.cam[0] / st.cam[2], y = st.cam[1] / st.cam[2];
    if (x * x + y * y > cam.largest_radius * cam.largest_radius) return false;
    double u, v, r, f;
    cam_project(cam, x, y, u, v, r, f);
    st.img[0] = u;
    st.img[1] = v;
    cam_derivs(cam, x, y, r, f, st.D);   // GetProjectionDerivs of THIS projection (no shared cache)
    if (r > cam.max_r) return true;
    if (u < 0 || v < 0 || u > cam.width || v > cam.height) return true;
    in_image = true;
    return true;
}

// CalcJacobian include/Tracker.h:125-136
__device__ __forceinline__ void td_jacobian(PoseState& st) {
    const double X = st.cam[0], Y = st.cam[1], Z = st.cam[2];
    const double iz = 1.0 / Z;
    // generator_field(m, (X,Y,Z,1)): m<3 -> e_m ; 3 -> (0,-Z,Y) ; 4 -> (Z,0,-X) ; 5 -> (-Y,X,0)
    const double gx[6] = {1, 0, 0, 0, Z, -Y};
    const double gy[6] = {0, 1, 0, -Z, 0, X};
    const double gz[6] = {0, 0, 1, Y, -X, 0};
#pragma unroll
    for (int m = 0; m < 6; m++) {
        const double mx = (gx[m] - X * gz[m] * iz) * iz;
        const double my = (gy[m] - Y * gz[m] * iz) * iz;
        st.J[m] = st.D[0] * mx + st.D[1] * my;
        st.J[6 + m] = st.D[2] * mx + st.D[3] * my;
    }
}

// (body of pose_gn_kernel / pose_gn_batch_kernel: a 1024-thread workgroup)
__device__ __forceinline__ void pose_gn_body(const DevCam& cam, int n, const ptam_pose_meas* __restrict__ meas,
                                             const ptam_projection* __restrict__ entry,
                                             double* __restrict__ pose_io, const ptam_gn_opts& opts,
                                             PoseState* __restrict__ st, int* __restrict__ flags,
                                             double* __restrict__ updates, const int* __restrict__ n_dev, const PoseIn* pin,
                                             const PoseChainIo& io, int size_guard) {
    __shared__ GnShared sh;
    const int tid = threadIdx.x;
    if (size_guard == 2 && *n_dev <= GS_LIMIT) return;   // the resident chain enqueues both kernels: the list length picks one
    if (n_dev) n = min(n, max(*n_dev, 0));   // counted variant: the measurement list was compacted on the device
    if (tid < 12) sh.pose[tid] = (pin && pin->use) ? pin->v[tid] : pose_io[tid];   // (a kernel argument: a local copy indexed by tid would be scratch memory)
    __syncthreads();
    for (int i = tid; i < n; i += GN_THREADS) {
        PoseState s;
        s.found = 1;
        s.pad_ = 0;
        s.e[0] = s.e[1] = s.e2 = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) s.J[k] = 0;
        if (entry) {
#pragma unroll
            for (int k = 0; k < 3; k++) s.cam[k] = entry[i].cam[k];
            s.img[0] = entry[i].image[0];
            s.img[1] = entry[i].image[1];
#pragma unroll
            for (int k = 0; k < 4; k++) s.D[k] = entry[i].derivs[k];
        } else {
            s.img[0] = s.img[1] = 0;
            s.D[0] = s.D[1] = s.D[2] = s.D[3] = 0;
            bool in_image;
            td_project(cam, sh.pose, meas[i].world, s, in_image);
            if (!in_image) s.found = 0;   // not in the potentially-visible set (src/Tracker.cc:456-458)
        }
        st[i] = s;
        if (flags) flags[i] = 0;
    }
    __syncthreads();
    const double* found = meas[0].found;
    const double* snoise = &meas[0].sqrt_inv_noise;
    const int stride = sizeof(ptam_pose_meas) / sizeof(double);
    for (int iter = 0; iter < opts.iterations; iter++) {
        const bool nonlinear = (opts.nonlinear_mask >> iter) & 1u;
        if (iter != 0) {
            if (nonlinear) {
                for (int i = tid; i < n; i += GN_THREADS)
                    if (st[i].found) {
                        PoseState s = st[i];
                        bool in_image;
                        if (!td_project(cam, sh.pose, meas[i].world, s, in_image))   // keeps img/D when not reached
                            atomicAdd(&g_pose_hazards, 1ull);                          // (counted: ptam_ctx_cache_hazards)
                        st[i] = s;
                    }
            } else {
                for (int i = tid; i < n; i += GN_THREADS)
                    if (st[i].found) {   // LinearUpdate include/Tracker.h:139-142
                        double a = 0, b = 0;
#pragma unroll
                        for (int m = 0; m < 6; m++) {
                            a += st[i].J[m] * sh.mu[m];
                            b += st[i].J[6 + m] * sh.mu[m];
                        }
                        st[i].img[0] += a;
                        st[i].img[1] += b;
                    }
            }
        }
        if (nonlinear)
            for (int i = tid; i < n; i += GN_THREADS)
                if (st[i].found) {
                    PoseState s = st[i];
                    td_jacobian(s);
                    st[i] = s;
                }
        __syncthreads();   // sh.mu (last update) fully consumed before it is overwritten
        const double ov = iter > opts.override_after ? opts.override_sigma_sq : 0.0;
        block_calc_pose_update(sh, st, n, found, snoise, stride, ov, opts.estimator, opts.prior,
                               iter == opts.mark_outliers_iter, flags);
        if (tid == 0) {
            double np[12];
            se3_exp_mul(sh.mu, sh.pose, np);   // mse3CamFromWorld = SE3<>::exp(v6Update) * mse3CamFromWorld
            for (int k = 0; k < 12; k++) sh.pose[k] = np[k];
            if (updates)
                for (int k = 0; k < 6; k++) updates[6 * iter + k] = sh.mu[k];
        }
        __syncthreads();
    }
    if (tid < 12) pose_io[tid] = sh.pose[tid];
    // resident chain: the measurements' TrackerData state goes back to the per-point table, scene depth sums
    if (io.td_base)
        for (int i = tid; i < n; i += GN_THREADS) {
            ptam_projection* o = (ptam_projection*)((char*)io.td_base + (size_t)(io.td_index ? io.td_index[i] : i) * io.td_stride);
#pragma unroll
            for (int k = 0; k < 3; k++) o->cam[k] = st[i].cam[k];
            o->image[0] = st[i].img[0];
            o->image[1] = st[i].img[1];
#pragma unroll
            for (int k = 0; k < 4; k++) o->derivs[k] = st[i].D[k];
        }
    if (io.depth_out) {
        double z1 = 0, z2 = 0;
        for (int i = tid; i < n; i += GN_THREADS) {
            const double z = st[i].cam[2];
            z1 += z;
            z2 += z * z;
        }
        z1 = wave_sum_f64(z1);
        z2 = wave_sum_f64(z2);
        __syncthreads();
        if ((tid & 63) == 0) {
            sh.red[tid >> 6][0] = z1;
            sh.red[tid >> 6][1] = z2;
        }
        __syncthreads();
        if (tid == 0) {
            double a = 0, b = 0;
            for (int w = 0; w < GN_WAVES; w++) {
                a += sh.red[w][0];
                b += sh.red[w][1];
            }
            io.depth_out[0] = a;
            io.depth_out[1] = b;
            io.depth_out[2] = (double)n;
            if (io.result_depth) {
                io.result_depth[0] = a;
                io.result_depth[1] = b;
                io.result_depth[2] = (double)n;
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
}

__global__ void __launch_bounds__(GN_THREADS) pose_gn_kernel(DevCam cam, int n, const ptam_pose_meas* __restrict__ meas,
                                                             const ptam_projection* __restrict__ entry,
                                                             double* __restrict__ pose_io, ptam_gn_opts opts,
                                                             PoseState* __restrict__ st, int* __restrict__ flags,
                                                             double* __restrict__ updates, const int* __restrict__ n_dev, PoseIn pin,
                                                             PoseChainIo io, int size_guard) {
    pose_gn_body(cam, n, meas, entry, pose_io, opts, st, flags, updates, n_dev, &pin, io, size_guard);
}
// one workgroup per frame of a batch (ptam_track_map_frames_batch): the same body on the frame's own lists
__global__ void __launch_bounds__(GN_THREADS) pose_gn_batch_kernel(DevCam cam, const PoseBatchItem* __restrict__ items, ptam_gn_opts opts,
                                                                   int size_guard) {
    const PoseBatchItem it = items[blockIdx.x];
    pose_gn_body(cam, it.n_cap, it.meas, it.entry, it.pose_io, opts, (PoseState*)it.st, it.flags, it.updates, it.n_dev, nullptr, it.io, size_guard);
}

template <int MPT, int THREADS>
__global__ void __launch_bounds__(THREADS) pose_gn_small_kernel(DevCam cam, int n, const ptam_pose_meas* __restrict__ meas,
                                                                   const ptam_projection* __restrict__ entry,
                                                                   double* __restrict__ pose_io, ptam_gn_opts opts,
                                                                   int* __restrict__ flags, double* __restrict__ updates,
                                                                   ulonglong2* __restrict__ host_slots, unsigned long long seq,
                                                                   const int* __restrict__ n_dev, PoseIn pin, PoseChainIo io, int size_guard) {
    PoseArrayLoader ld{meas, entry, n_dev};
    pose_gn_small_body<MPT, THREADS>(cam, n, ld, pose_io, opts, flags, updates, host_slots, seq, n_dev, &pin, io, size_guard);
}
template <int MPT, int THREADS>
__global__ void __launch_bounds__(THREADS) pose_gn_small_batch_kernel(DevCam cam, const PoseBatchItem* __restrict__ items, ptam_gn_opts opts,
                                                                         int size_guard) {
    const PoseBatchItem it = items[blockIdx.x];
    PoseArrayLoader ld{it.meas, it.entry, it.n_dev};
    pose_gn_small_body<MPT, THREADS>(cam, min(it.n_cap, THREADS * MPT), ld, it.pose_io, opts, it.flags, it.updates, nullptr, 0ull,
                                     it.n_dev, nullptr, it.io, size_guard);
}

// instantiation and workgroup size for a list of at most n_cap (<= GS_LIMIT) measurements: ONE wave for the coarse set's
// sizes (<= 64: every barrier of the kernel is then a no-op and the 27 sums meet after a single DPP step), one measurement
// per thread up to 256, four up to 1024
typedef void (*pose_small_fn)(DevCam, int, const ptam_pose_meas*, const ptam_projection*, double*, ptam_gn_opts, int*, double*, ulonglong2*,
                              unsigned long long, const int*, PoseIn, PoseChainIo, int);
static pose_small_fn pose_small_pick(int n_cap, int* threads) {
    static const bool no_wave = ptam_ab_env("PTAM_POSE_NO_WAVE") != nullptr;   // (A/B runs)
    if (n_cap <= GS_WAVE_LIMIT && !no_wave) {
        *threads = GS_WAVE_LIMIT;
        return pose_gn_small_kernel<1, GS_WAVE_LIMIT>;
    }
    *threads = GS_THREADS;
    return n_cap <= GS_THREADS ? pose_gn_small_kernel<1, GS_THREADS> : pose_gn_small_kernel<GS_MPT, GS_THREADS>;
}

__global__ void __launch_bounds__(GN_THREADS) calc_pose_update_kernel(int n, const ptam_pose_update_meas* __restrict__ meas,
                                                                      double override_sigma, int est, double prior,
                                                                      PoseState* __restrict__ st, int* __restrict__ flags,
                                                                      double* __restrict__ mu_out) {
    __shared__ GnShared sh;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += GN_THREADS) {
        PoseState s;
        s.found = 1;
        s.pad_ = 0;
        s.cam[0] = s.cam[1] = s.cam[2] = 0;
        s.D[0] = s.D[1] = s.D[2] = s.D[3] = 0;
        s.img[0] = meas[i].image[0];
        s.img[1] = meas[i].image[1];
#pragma unroll
        for (int k = 0; k < 12; k++) s.J[k] = meas[i].jac[k];
        s.e[0] = s.e[1] = s.e2 = 0;
        st[i] = s;
        if (flags) flags[i] = 0;
    }
    __syncthreads();
    const int stride = sizeof(ptam_pose_update_meas) / sizeof(double);
    block_calc_pose_update(sh, st, n, meas[0].found, &meas[0].sqrt_inv_noise, stride, override_sigma, est, prior,
                           flags != nullptr, flags);
    if (tid < 6) mu_out[tid] = sh.mu[tid];
}

extern "C" {

void ptam_gn_opts_default(ptam_gn_opts* o) {
    if (!o) return;
    o->iterations = 10;
    o->nonlinear_mask = 0x211;   // iter 0, 4, 9  src/Tracker.cc:618
    o->override_after = 5;       // :637
    o->override_sigma_sq = 16.0;
    o->mark_outliers_iter = 9;   // :640
    o->estimator = PTAM_EST_TUKEY;
    o->prior = 100.0;            // :974
}

static int pose_gn_host(ptam_ctx* ctx, int n, const ptam_pose_meas* meas, const ptam_projection* entry,
                        double pose_inout[12], const ptam_gn_opts* opts, int32_t* outlier_flags, double* updates_out,
                        ptam_projection* state_out) {
    ARG_TRY(ctx && n >= 0 && pose_inout);
    ptam_gn_opts o;
    if (opts)
        o = *opts;
    else
        ptam_gn_opts_default(&o);
    ARG_TRY(o.iterations >= 0 && o.iterations <= 32);
    if (n == 0) {
        // CalcPoseUpdate returns a zero update for an empty set (:955-956): pose unchanged
        if (updates_out) std::memset(updates_out, 0, sizeof(double) * 6 * o.iterations);
        return PTAM_OK;
    }
    ARG_TRY(meas);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bm = (size_t)n * sizeof(ptam_pose_meas), be = entry ? (size_t)n * sizeof(ptam_projection) : 0,
                 bs = (size_t)n * sizeof(PoseState), bf = (size_t)n * 4, bu = (size_t)6 * 32 * 8;
    const size_t b_in = bm + be + 96;   // [measurements | entry state | pose]: ONE upload
    const size_t b_so = state_out ? (size_t)n * sizeof(ptam_projection) : 0;
    void* s;
    int rc = ctx_scratch(ctx, b_in + bs + bf + bu + b_so + 128, &s);
    if (rc) return rc;
    PoseChainIo io{};
    if (state_out) {   // the measurements' TrackerData state at loop exit (entry state of a following loop, src/Tracker.cc:617)
        io.td_base = (char*)s + ((b_in + bs + bf + bu + 63) & ~(size_t)63);
        io.td_index = nullptr;
        io.td_stride = (int)sizeof(ptam_projection);
        HIP_TRY(hipMemsetAsync(io.td_base, 0, b_so, ctx->stream));
    }
    char* p = (char*)s;
    ptam_pose_meas* d_m = (ptam_pose_meas*)p;
    ptam_projection* d_e = entry ? (ptam_projection*)(p + bm) : nullptr;
    double* d_pose = (double*)(p + bm + be);
    p += b_in;
    PoseState* d_s = (PoseState*)p;
    p += bs;
    double* d_u = (double*)p;
    p += bu;
    int* d_f = (int*)p;
    // host buffers are pageable: everything goes through the context's pinned staging so that no copy blocks
    // pinned layout: [inputs b_in][12 result slots of 16 B][flags][updates]
    const bool small = n <= GS_THREADS * GS_MPT;
    const size_t b_slots = 12 * 16, b_upd = sizeof(double) * 6 * o.iterations;
    void* pin;
    rc = ctx_pinned(ctx, b_in + b_slots + bf + b_upd + 64, &pin);
    if (rc) return rc;
    char* hp = (char*)pin;
    std::memcpy(hp, meas, bm);
    if (entry) std::memcpy(hp + bm, entry, be);
    std::memcpy(hp + bm + be, pose_inout, 96);
    const size_t o_slots = (b_in + 15) & ~(size_t)15;
    volatile unsigned long long* slots = (volatile unsigned long long*)(hp + o_slots);
    char* hf = hp + o_slots + b_slots;
    char* hu = hf + bf;
    for (int i = 0; i < 12; i++) slots[2 * i + 1] = 0;   // (the staging buffer is shared: no stale sequence numbers)
    HIP_TRY(hipMemcpyAsync(d_m, hp, b_in, hipMemcpyHostToDevice, ctx->stream));
    const unsigned long long seq = ++ctx->pose_seq;
    if (small) {
        int thr;
        const pose_small_fn fn = pose_small_pick(n, &thr);
        hipLaunchKernelGGL(fn, dim3(1), dim3(thr), 0, ctx->stream,
                           ctx->cam, n, d_m, d_e, d_pose, o, d_f, d_u, (ulonglong2*)((char*)ctx->d_pinned + o_slots), seq,
                           (const int*)nullptr, PoseIn{}, io, 0);
    }
    else
        hipLaunchKernelGGL(pose_gn_kernel, dim3(1), dim3(GN_THREADS), 0, ctx->stream, ctx->cam, n, d_m, d_e, d_pose, o,
                           d_s, d_f, d_u, (const int*)nullptr, PoseIn{}, io, 0);
    HIP_TRY(hipGetLastError());
    const bool extras = outlier_flags || updates_out || state_out;
    if (outlier_flags) HIP_TRY(hipMemcpyAsync(hf, d_f, bf, hipMemcpyDeviceToHost, ctx->stream));
    if (updates_out) HIP_TRY(hipMemcpyAsync(hu, d_u, b_upd, hipMemcpyDeviceToHost, ctx->stream));
    if (small && !extras) {
        // only the pose is wanted: spin on the twelve (word, sequence) pairs the kernel writes into host-mapped memory
        auto arrived = [&]() {
            for (int i = 0; i < 12; i++)
                if (slots[2 * i + 1] != seq) return false;
            return true;
        };
        unsigned spins = 0;
        while (!arrived()) {
            if (++spins == 100000) {
                spins = 0;
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return PTAM_E_HIP;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int i = 0; i < 12; i++) {
            const unsigned long long w = slots[2 * i];
            std::memcpy(&pose_inout[i], &w, 8);
        }
        return PTAM_OK;
    }
    if (!small) HIP_TRY(hipMemcpyAsync(hp + o_slots, d_pose, 96, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    if (small) {
        for (int i = 0; i < 12; i++) {
            const unsigned long long w = slots[2 * i];
            std::memcpy(&pose_inout[i], &w, 8);
        }
    } else
        std::memcpy(pose_inout, hp + o_slots, 96);
    if (outlier_flags) std::memcpy(outlier_flags, hf, bf);
    if (updates_out) std::memcpy(updates_out, hu, b_upd);
    if (state_out) HIP_TRY(hipMemcpy(state_out, io.td_base, b_so, hipMemcpyDeviceToHost));   // (tests and the shim's staged path only)
    return PTAM_OK;
}

int ptam_pose_gn(ptam_ctx* ctx, int n, const ptam_pose_meas* meas, const ptam_projection* entry,
                 double pose_inout[12], const ptam_gn_opts* opts, int32_t* outlier_flags, double* updates_out) {
    return pose_gn_host(ctx, n, meas, entry, pose_inout, opts, outlier_flags, updates_out, nullptr);
}
int ptam_pose_gn_state(ptam_ctx* ctx, int n, const ptam_pose_meas* meas, const ptam_projection* entry,
                       double pose_inout[12], const ptam_gn_opts* opts, int32_t* outlier_flags, double* updates_out,
                       ptam_projection* state_out) {
    ARG_TRY(n == 0 || state_out);
    return pose_gn_host(ctx, n, meas, entry, pose_inout, opts, outlier_flags, updates_out, state_out);
}

static int pose_gn_dev_impl(ptam_ctx* ctx, int n, const int32_t* d_n, const ptam_pose_meas* d_meas, const ptam_projection* d_entry,
                            double* d_pose_inout, const ptam_gn_opts* opts, int32_t* d_outlier_flags, double* d_updates,
                            const double* pose_host_in, double* pose_host_out) {
    ARG_TRY(ctx && n >= 1 && d_meas && d_pose_inout);
    PoseIn pin{};
    if (pose_host_in) {
        std::memcpy(pin.v, pose_host_in, 96);
        pin.use = 1;
    }
    ptam_gn_opts o;
    if (opts)
        o = *opts;
    else
        ptam_gn_opts_default(&o);
    ARG_TRY(o.iterations >= 0 && o.iterations <= 32);
    HIP_TRY(hipSetDevice(ctx->device));
    // the kernels write the per-iteration updates unconditionally: give them the context's scratch when the caller
    // does not want them (the general kernel also keeps its per-measurement state there)
    const bool small = n <= GS_THREADS * GS_MPT;
    const size_t bs = small ? 0 : (size_t)n * sizeof(PoseState), bu = (size_t)6 * 32 * 8;
    void* s;
    int rc = ctx_scratch(ctx, bs + bu + 64, &s);
    if (rc) return rc;
    PoseState* d_s = (PoseState*)s;
    double* d_u = d_updates ? d_updates : (double*)((char*)s + bs);
    // pose_host_out: the fast kernel publishes the pose into host-mapped memory as (word, sequence) pairs
    volatile unsigned long long* slots = nullptr;
    ulonglong2* d_slots = nullptr;
    unsigned long long seq = 0;
    if (pose_host_out) {
        void* pin;
        rc = ctx_pinned(ctx, 12 * 16 + 64, &pin);
        if (rc) return rc;
        slots = (volatile unsigned long long*)pin;
        d_slots = (ulonglong2*)ctx->d_pinned;
        for (int i = 0; i < 12; i++) slots[2 * i + 1] = 0;
        seq = ++ctx->pose_seq;
    }
    if (small) {
        int thr;
        const pose_small_fn fn = pose_small_pick(n, &thr);
        hipLaunchKernelGGL(fn, dim3(1), dim3(thr), 0, ctx->stream,
                           ctx->cam, n, d_meas, d_entry, d_pose_inout, o, d_outlier_flags, d_u, d_slots, seq, (const int*)d_n, pin,
                           PoseChainIo{}, 0);
    }
    else
        hipLaunchKernelGGL(pose_gn_kernel, dim3(1), dim3(GN_THREADS), 0, ctx->stream, ctx->cam, n, d_meas, d_entry, d_pose_inout, o,
                           d_s, d_outlier_flags, d_u, (const int*)d_n, pin, PoseChainIo{}, 0);
    HIP_TRY(hipGetLastError());
    if (!pose_host_out) return PTAM_OK;
    if (!small) {
        HIP_TRY(hipMemcpyAsync((void*)slots, d_pose_inout, 96, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ptam_stream_wait(ctx->stream));
        std::memcpy(pose_host_out, (const void*)slots, 96);
        return PTAM_OK;
    }
    auto arrived = [&]() {
        for (int i = 0; i < 12; i++)
            if (slots[2 * i + 1] != seq) return false;
        return true;
    };
    unsigned spins = 0;
    while (!arrived()) {
        if (++spins == 100000) {
            spins = 0;
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q != hipSuccess && q != hipErrorNotReady) return PTAM_E_HIP;
            if (q == hipSuccess && !arrived()) return PTAM_E_HIP;   // the stream drained without the kernel publishing
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < 12; i++) {
        const unsigned long long w = slots[2 * i];
        std::memcpy(&pose_host_out[i], &w, 8);
    }
    return PTAM_OK;
}

}   // extern "C"

// resident TrackMap chain: the list length sits in device memory and may exceed what the register-resident kernel holds,
// so BOTH kernels are enqueued and the length picks the one that runs (the other leaves at once)
// mode 0: the register-resident kernel and, when the list may exceed it, the general one behind it (the length picks one on the
// device).  mode 1: the register-resident kernel only — on a longer list it publishes io.seq with POSE_CHAIN_LONG set instead
// of a result, and the caller comes back with mode 2: the general kernel alone.  (A frame's list is longer than 1024 only in
// maps that put more than a thousand points into one view; the general kernel launched behind every frame "just in case" and
// returning at once cost the single-camera chain 4.3 us per frame.)
int pose_launch_chain(ptam_ctx* ctx, int n_cap, const int* d_n, const ptam_pose_meas* d_meas, const ptam_projection* d_entry,
                      double* d_pose_inout, const ptam_gn_opts* opts, int32_t* d_outlier_flags, const PoseChainIo& io, int mode) {
    ARG_TRY(ctx && n_cap >= 1 && d_n && d_meas && d_pose_inout && opts);
    const bool may_be_long = n_cap > GS_LIMIT;
    const size_t bs = may_be_long ? (size_t)n_cap * sizeof(PoseState) : 0, bu = (size_t)6 * 32 * 8;
    void* s;
    int rc = ctx_scratch(ctx, bs + bu + 64, &s);
    if (rc) return rc;
    double* d_u = (double*)((char*)s + bs);
    // (a list that cannot exceed 256 entries — the coarse set — runs with one measurement per thread: a quarter of the
    //  straight-line work per iteration and a ranking select without a histogram)
    int thr;
    const pose_small_fn fn = pose_small_pick(std::min(n_cap, GS_LIMIT), &thr);
    if (mode != 2)
        hipLaunchKernelGGL(fn, dim3(1), dim3(thr), 0, ctx->stream,
                           ctx->cam, std::min(n_cap, GS_LIMIT), d_meas, d_entry, d_pose_inout, *opts, d_outlier_flags, d_u, (ulonglong2*)nullptr,
                           0ull, d_n, PoseIn{}, io, may_be_long ? (mode == 1 ? 3 : 1) : 0);
    if (may_be_long && mode != 1)
        hipLaunchKernelGGL(pose_gn_kernel, dim3(1), dim3(GN_THREADS), 0, ctx->stream, ctx->cam, n_cap, d_meas, d_entry, d_pose_inout, *opts,
                           (PoseState*)s, d_outlier_flags, d_u, d_n, PoseIn{}, io, mode == 2 ? 0 : 2);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

extern "C" {

int ptam_pose_gn_dev(ptam_ctx* ctx, int n, const ptam_pose_meas* d_meas, const ptam_projection* d_entry, double* d_pose_inout,
                     const ptam_gn_opts* opts, int32_t* d_outlier_flags, double* d_updates) {
    return pose_gn_dev_impl(ctx, n, nullptr, d_meas, d_entry, d_pose_inout, opts, d_outlier_flags, d_updates, nullptr, nullptr);
}

int ptam_pose_gn_dev_counted(ptam_ctx* ctx, int n_cap, const int32_t* d_n, const ptam_pose_meas* d_meas,
                             const ptam_projection* d_entry, double* d_pose_inout, const ptam_gn_opts* opts,
                             int32_t* d_outlier_flags, double* d_updates, const double* pose_host_in, double* pose_host_out) {
    ARG_TRY(d_n);
    return pose_gn_dev_impl(ctx, n_cap, d_n, d_meas, d_entry, d_pose_inout, opts, d_outlier_flags, d_updates, pose_host_in,
                            pose_host_out);
}

// ---- Tracker::SearchForPoints' bookkeeping for a batch (src/Tracker.cc:883-909): every patch that was found (and, when
// sub-pixel results are given, whose refinement converged :898-904) becomes one measurement of the pose solve,
// {v3WorldPos, v2Found = coarse / sub-pixel position, dSqrtInvNoise = 1 / LevelScale(level)}, in query order.
// One workgroup, stable compaction: per 1024-slice a ballot per wave, wave offsets through LDS.
__global__ void __launch_bounds__(1024) gather_pose_meas_kernel(int n, const ptam_patch_query* __restrict__ q,
                                                                const ptam_patch_result* __restrict__ res,
                                                                const ptam_subpix_result* __restrict__ sub,
                                                                const char* __restrict__ world, int world_stride,
                                                                ptam_pose_meas* __restrict__ out, int* __restrict__ src_index,
                                                                int* __restrict__ count, int* __restrict__ level_found) {
    __shared__ int wcnt[16];
    __shared__ int base_s;
    __shared__ int lvl[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base_s = 0;
    if (tid < 4) lvl[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        bool ok = false;
        int level = 0;
        double fx = 0, fy = 0;
        if (i < n) {
            level = q[i].level;
            ok = level >= 0 && res[i].found != 0;
            fx = res[i].pos[0];
            fy = res[i].pos[1];
            if (ok && sub) {
                ok = sub[i].converged != 0;
                fx = sub[i].pos[0];
                fy = sub[i].pos[1];
            }
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) wcnt[wid] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wid; w++) off += wcnt[w];
        if (ok) {
            const int o = off + __popcll(m & ((1ull << lane) - 1ull));
            const double* wp = (const double*)(world + (size_t)i * world_stride);
            out[o].world[0] = wp[0];
            out[o].world[1] = wp[1];
            out[o].world[2] = wp[2];
            out[o].found[0] = fx;
            out[o].found[1] = fy;
            out[o].sqrt_inv_noise = 1.0 / (double)(1 << level);
            if (src_index) src_index[o] = i;
            if (level < 4) atomicAdd(&lvl[level], 1);   // manMeasFound :892
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; w++) t += wcnt[w];
            base_s += t;
        }
        __syncthreads();
    }
    if (tid == 0) *count = base_s;
    if (level_found && tid < 4) level_found[tid] = lvl[tid];
}

int ptam_gather_pose_meas_dev(ptam_ctx* ctx, int n, const ptam_patch_query* d_queries, const ptam_patch_result* d_results,
                              const ptam_subpix_result* d_subpix, const void* d_world, int world_stride_bytes,
                              ptam_pose_meas* d_meas_out, int32_t* d_src_index, int32_t* d_count, int32_t* d_level_found) {
    ARG_TRY(ctx && n >= 0 && d_count && (n == 0 || (d_queries && d_results && d_world && d_meas_out)));
    ARG_TRY(world_stride_bytes >= 24 && world_stride_bytes % 8 == 0);
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(gather_pose_meas_kernel, dim3(1), dim3(1024), 0, ctx->stream, n, d_queries, d_results, d_subpix,
                       (const char*)d_world, world_stride_bytes, d_meas_out, d_src_index, d_count, d_level_found);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

int ptam_calc_pose_update(ptam_ctx* ctx, int n, const ptam_pose_update_meas* meas, double override_sigma_sq,
                          int estimator, double prior, double mu_out[6], int32_t* weight_zero_flags) {
    ARG_TRY(ctx && n >= 0 && mu_out);
    if (n == 0) {
        for (int i = 0; i < 6; i++) mu_out[i] = 0;
        return PTAM_OK;
    }
    ARG_TRY(meas);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bm = (size_t)n * sizeof(ptam_pose_update_meas), bs = (size_t)n * sizeof(PoseState), bf = (size_t)n * 4;
    void* s;
    int rc = ctx_scratch(ctx, bm + bs + bf + 64, &s);
    if (rc) return rc;
    char* p = (char*)s;
    ptam_pose_update_meas* d_m = (ptam_pose_update_meas*)p;
    p += bm;
    PoseState* d_s = (PoseState*)p;
    p += bs;
    double* d_mu = (double*)p;
    p += 64;
    int* d_f = (int*)p;
    HIP_TRY(hipMemcpyAsync(d_m, meas, bm, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(calc_pose_update_kernel, dim3(1), dim3(GN_THREADS), 0, ctx->stream, n, d_m, override_sigma_sq,
                       estimator, prior, d_s, weight_zero_flags ? d_f : nullptr, d_mu);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mu_out, d_mu, 48, hipMemcpyDeviceToHost, ctx->stream));
    if (weight_zero_flags) HIP_TRY(hipMemcpyAsync(weight_zero_flags, d_f, bf, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

}   // extern "C"

// every kernel of this file resolved once, when a context is created: the first launch of a kernel otherwise pays for
// loading the code object / resolving the function — 10-28 ms in the middle of the first frame or the first adjustment
// the same for nb frames at once: one workgroup per frame (d_items: nb PoseBatchItem in device memory); n_cap_max = the largest
// capacity among them picks the instantiation
int pose_launch_chain_batch(ptam_ctx* ctx, int nb, int n_cap_max, const PoseBatchItem* d_items, const ptam_gn_opts* opts) {
    ARG_TRY(ctx && nb >= 1 && n_cap_max >= 1 && d_items && opts);
    const bool may_be_long = n_cap_max > GS_LIMIT;
    const int cap = std::min(n_cap_max, GS_LIMIT);
    if (cap <= GS_WAVE_LIMIT)
        hipLaunchKernelGGL((pose_gn_small_batch_kernel<1, GS_WAVE_LIMIT>), dim3(nb), dim3(GS_WAVE_LIMIT), 0, ctx->stream, ctx->cam, d_items, *opts, may_be_long ? 1 : 0);
    else if (cap <= GS_THREADS)
        hipLaunchKernelGGL((pose_gn_small_batch_kernel<1, GS_THREADS>), dim3(nb), dim3(GS_THREADS), 0, ctx->stream, ctx->cam, d_items, *opts, may_be_long ? 1 : 0);
    else
        hipLaunchKernelGGL((pose_gn_small_batch_kernel<GS_MPT, GS_THREADS>), dim3(nb), dim3(GS_THREADS), 0, ctx->stream, ctx->cam, d_items, *opts, may_be_long ? 1 : 0);
    if (may_be_long) hipLaunchKernelGGL(pose_gn_batch_kernel, dim3(nb), dim3(GN_THREADS), 0, ctx->stream, ctx->cam, d_items, *opts, 2);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}
// scratch of one frame's pose loops inside the context's scratch buffer (what pose_launch_chain takes for itself)
int pose_chain_scratch(ptam_ctx* ctx, int n_cap, void** st_out, double** updates_out) {
    const bool may_be_long = n_cap > GS_LIMIT;
    const size_t bs = may_be_long ? (size_t)n_cap * sizeof(PoseState) : 0, bu = (size_t)6 * 32 * 8;
    void* s;
    const int rc = ctx_scratch(ctx, bs + bu + 64, &s);
    if (rc) return rc;
    *st_out = s;
    *updates_out = (double*)((char*)s + bs);
    return PTAM_OK;
}

int pose_hazards_read_pose(hipStream_t st, unsigned long long* out) {   // (this translation unit's copy of the counter)
    HIP_TRY(hipMemcpyFromSymbolAsync(out, HIP_SYMBOL(g_pose_hazards), 8, 0, hipMemcpyDeviceToHost, st));
    return PTAM_OK;
}
void pose_preload_kernels() {
    ptam_preload((const void*)pose_gn_kernel);
    ptam_preload((const void*)pose_gn_small_kernel<1, GS_THREADS>);
    ptam_preload((const void*)pose_gn_small_kernel<1, GS_WAVE_LIMIT>);
    ptam_preload((const void*)pose_gn_small_kernel<GS_MPT, GS_THREADS>);
    ptam_preload((const void*)pose_gn_small_batch_kernel<1, GS_WAVE_LIMIT>);
    ptam_preload((const void*)pose_gn_small_batch_kernel<1, GS_THREADS>);
    ptam_preload((const void*)pose_gn_small_batch_kernel<GS_MPT, GS_THREADS>);
    ptam_preload((const void*)pose_gn_batch_kernel);
    ptam_preload((const void*)calc_pose_update_kernel);
    ptam_preload((const void*)gather_pose_meas_kernel);
}
