// common.h — shared host/device definitions of libptam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptam_hip.h"

// ---- error plumbing -------------------------------------------------------------------------
void ptam_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            ptam_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                           __LINE__);                                                        \
            return PTAM_E_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define ARG_TRY(cond)                                                        \
    do {                                                                     \
        if (!(cond)) {                                                       \
            ptam_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
            return PTAM_E_ARG;                                               \
        }                                                                    \
    } while (0)

// ---- camera model on the device (pure function of these constants; no cached state) ----------
// RefreshParams src/ATANCamera.cc:27-66
struct DevCam {
    double fx, fy, cx, cy;        // mvFocal, mvCenter (pixels)
    double w, two_tan, w_inv, dist_enabled;
    double largest_radius, max_r;
    double width, height;
    double inv_fx, inv_fy, one_over_two_tan;   // mvInvFocal, mdOneOver2Tan (UnProject)
    double one_pixel_dist;                     // mdOnePixelDist
};

struct ptam_ctx {
    int device;
    hipStream_t stream;
    hipStream_t stream_alt;   // second queue, created when a bundle's rejected trial first needs it (bundle.hip: QueueTurn)
    ptam_cam_params params;
    DevCam cam;
    int halfsample;
    // grow-on-demand device scratch + pinned host staging (owned by this ctx / thread)
    void* d_scratch;
    size_t d_scratch_cap;
    void* h_pinned;
    void* d_pinned;          // device address of h_pinned (host-mapped): kernels publish small results into it
    size_t h_pinned_cap;
    unsigned long long pose_seq;
    // Released bundle memory is kept for the next bundle of this context (MapMaker builds a new Bundle for every
    // adjustment, src/MapMaker.cc:838-845): hipMalloc / hipFree / hipHostMalloc around queued work cost far more than their
    // own time here — mapping memory into the GPU's address space makes the driver evict and restore the queues (10-28 ms).
    struct Cached {
        void* p;
        size_t bytes;
    };
    Cached dev_cache[8];    // device blocks (a bundle holds three: its main block, its Schur work lists, its measurements as they were added)
    Cached host_cache[2];   // host-mapped mailboxes
    Cached pin_cache[12];   // pinned host arrays of released bundles (the measurements as they are added: PinVec, bundle.hip)
    unsigned* d_smap;       // the Schur tile kernel's index maps (schur_index_map_device: a compile-time constant, uploaded once)
    int n_cu;               // compute units of the device
};
#define CTX_NCACHE(a) ((int)(sizeof(a) / sizeof((a)[0])))
int ctx_cache_take(ptam_ctx::Cached* c, int slots, size_t bytes, void** out, size_t* cap);   // smallest cached block >= bytes, or null
void* ctx_cache_give(ptam_ctx::Cached* c, int slots, void* p, size_t bytes);                 // returns the pointer the caller must free (or null)

// hipStreamSynchronize sleeps on an interrupt, and waking from it was measured at up to 7 ms on this platform; the waits
// of this library end within microseconds to a few milliseconds, so every one of them polls first (2 ms) and only then sleeps.
hipError_t ptam_stream_wait(hipStream_t stream);
void ptam_preload(const void* kernel);   // hipFuncGetAttributes: forces the kernel's code object to be loaded
void ba_preload_kernels();
void solve_preload_kernels();
void pose_preload_kernels();
void patch_preload_kernels();
void kf_preload_kernels();
void pvs_preload_kernels();
void trackmap_preload_kernels();
int pose_hazards_read_pose(hipStream_t st, unsigned long long* out);
int pose_hazards_read_trackmap(hipStream_t st, unsigned long long* out);
int ctx_scratch(ptam_ctx* ctx, size_t bytes, void** out);     // device scratch >= bytes
int ctx_pinned(ptam_ctx* ctx, size_t bytes, void** out);      // pinned host staging >= bytes

// Measurement switches (kernel shapes, work splits, forms of a stage side by side) are read from the environment only in the
// instrumented builds of tools/ (-DPTAM_AB_SWITCHES: `make ab` -> tools/_ab/libptam_hip.so); the product library has none of them.
// What the product does read: PTAM_LDLT_NO_CHAIN (launch-per-block-column camera solve only), PTAM_CH_SPIN_LIMIT (how long a
// workgroup of the persistent solve waits for another one) and PTAM_TWO_QUEUES (a second queue for a rejected trial's continuation) — operating switches, documented in ptam_hip.h — and the PTAM_DEBUG_*
// diagnostics, which only print.
#ifdef PTAM_AB_SWITCHES
#define ptam_ab_env(name) getenv(name)
#else
#define ptam_ab_env(name) ((const char*)nullptr)
#endif

// ---- small device math ---------------------------------------------------------------------
#define PTAM_HD __host__ __device__ __forceinline__

// rtrans_factor include/ATANCamera.h:143-149 + Project src/ATANCamera.cc:109-121
#ifdef __HIPCC__
#include "atan_cr.h"
#endif
// (cam_project, cam_derivs, se3_apply: no FMA contraction — which product of a * b + c * d gets fused depends on the kernel a copy is
//  inlined into, and these feed truncations: ir() of the image position, the grey levels of a warped template through the warp
//  matrix.  Plain products and sums are what the reference's compiler emits: pvs_device.h)
PTAM_HD void cam_project(const DevCam& c, double x, double y, double& u, double& v, double& r, double& f) {
#pragma clang fp contract(off)
    r = sqrt(x * x + y * y);
#ifdef __HIP_DEVICE_COMPILE__
    // (atan_cr.h: correctly rounded — OCML's atan differs from a correctly rounded one in 15 % of its results, the host libm's in 0.1 %)
    f = (r < 0.001 || c.w == 0.0) ? 1.0 : (c.w_inv * atan_cr(r * c.two_tan) / r);
#else
    f = (r < 0.001 || c.w == 0.0) ? 1.0 : (c.w_inv * atan(r * c.two_tan) / r);
#endif
    u = c.cx + c.fx * (f * x);
    v = c.cy + c.fy * (f * y);
}
// GetProjectionDerivs src/ATANCamera.cc:179-209 for the projection (x, y, r, f)
PTAM_HD void cam_derivs(const DevCam& c, double x, double y, double r_in, double f, double D[4]) {
#pragma clang fp contract(off)
    const double k = c.two_tan;
    const double r = r_in * c.dist_enabled;
    double dx, dy;
    if (r < 0.01) {
        dx = 0.0;
        dy = 0.0;
    } else {
        const double den = r * r * (1 + k * k * r * r);
        dx = c.w_inv * (k * x) / den - x * f / (r * r);
        dy = c.w_inv * (k * y) / den - y * f / (r * r);
    }
    D[0] = c.fx * (dx * x + f);
    D[2] = c.fy * (dx * y);
    D[1] = c.fx * (dy * x);
    D[3] = c.fy * (dy * y + f);
}

// pose = R row-major (9) + t (3)
PTAM_HD void se3_apply(const double* T, double x, double y, double z, double& ox, double& oy, double& oz) {
#pragma clang fp contract(off)
    ox = T[9] + (T[0] * x + T[1] * y + T[2] * z);
    oy = T[10] + (T[3] * x + T[4] * y + T[5] * z);
    oz = T[11] + (T[6] * x + T[7] * y + T[8] * z);
}

// TooN SE3<>::exp: the rotation R (row-major) and the translation et of exp(mu)
template <bool SERIES = false>
PTAM_HD void se3_exp_parts(const double* mu, double* R, double* et) {
    const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double tx = mu[0], ty = mu[1], tz = mu[2], wx = mu[3], wy = mu[4], wz = mu[5];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    const double cx = wy * tz - wz * ty, cy = wz * tx - wx * tz, cz = wx * ty - wy * tx;
    double A, B;
    if (theta_sq < 1e-8) {
        A = 1.0 - one_6th * theta_sq;
        B = 0.5;
        et[0] = tx + 0.5 * cx;
        et[1] = ty + 0.5 * cy;
        et[2] = tz + 0.5 * cz;
    } else {
        double Cc;
        if (theta_sq < 1e-6) {
            Cc = one_6th * (1.0 - one_20th * theta_sq);
            A = 1.0 - theta_sq * Cc;
            B = 0.5 - 0.25 * one_6th * theta_sq;
        } else if (SERIES && theta_sq < 0.25) {
            // device fast path (pose solve: one thread runs this ten times per frame): the three coefficients as even
            // series in theta^2 — A = sum (-t)^k/(2k+1)!, B = sum (-t)^k/(2k+2)!, C = sum (-t)^k/(2k+3)! — nine terms,
            // truncation < 1e-19 for theta^2 < 1/4; no sqrt, sin, cos or division (a couple of ulp from the libm route)
            const double t = -theta_sq;
            A = 1.0 + t * (1.0 / 6 + t * (1.0 / 120 + t * (1.0 / 5040 + t * (1.0 / 362880 + t * (1.0 / 39916800 + t * (1.0 / 6227020800.0 +
                t * (1.0 / 1307674368000.0 + t * (1.0 / 355687428096000.0))))))));
            B = 0.5 + t * (1.0 / 24 + t * (1.0 / 720 + t * (1.0 / 40320 + t * (1.0 / 3628800 + t * (1.0 / 479001600 + t * (1.0 / 87178291200.0 +
                t * (1.0 / 20922789888000.0 + t * (1.0 / 6402373705728000.0))))))));
            Cc = 1.0 / 6 + t * (1.0 / 120 + t * (1.0 / 5040 + t * (1.0 / 362880 + t * (1.0 / 39916800 + t * (1.0 / 6227020800.0 +
                 t * (1.0 / 1307674368000.0 + t * (1.0 / 355687428096000.0 + t * (1.0 / 121645100408832000.0))))))));
        } else {
            const double theta = sqrt(theta_sq);
            const double inv_theta = 1.0 / theta;
            A = sin(theta) * inv_theta;
            B = (1 - cos(theta)) * (inv_theta * inv_theta);
            Cc = (1 - A) * (inv_theta * inv_theta);
        }
        const double dx = wy * cz - wz * cy, dy = wz * cx - wx * cz, dz = wx * cy - wy * cx;
        et[0] = tx + B * cx + Cc * dx;
        et[1] = ty + B * cy + Cc * dy;
        et[2] = tz + B * cz + Cc * dz;
    }
    {
        const double wx2 = wx * wx, wy2 = wy * wy, wz2 = wz * wz;
        R[0] = 1.0 - B * (wy2 + wz2);
        R[4] = 1.0 - B * (wx2 + wz2);
        R[8] = 1.0 - B * (wx2 + wy2);
        double a = A * wz, b = B * (wx * wy);
        R[1] = b - a;
        R[3] = b + a;
        a = A * wy;
        b = B * (wx * wz);
        R[2] = b + a;
        R[6] = b - a;
        a = A * wx;
        b = B * (wy * wz);
        R[5] = b - a;
        R[7] = b + a;
    }
}
// ... followed by left-multiplication: out = exp(mu) * T
template <bool SERIES = false>
PTAM_HD void se3_exp_mul(const double* mu, const double* T, double* out) {
    double R[9], et[3];
    se3_exp_parts<SERIES>(mu, R, et);
    double o[12];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            o[r * 3 + c] = R[r * 3 + 0] * T[0 * 3 + c] + R[r * 3 + 1] * T[1 * 3 + c] + R[r * 3 + 2] * T[2 * 3 + c];
        o[9 + r] = et[r] + (R[r * 3 + 0] * T[9] + R[r * 3 + 1] * T[10] + R[r * 3 + 2] * T[11]);
    }
    for (int i = 0; i < 12; i++) out[i] = o[i];
}

// M-estimators include/Tools.h:128-228
PTAM_HD double est_sqrt_weight(int est, double e2, double s2) {
    if (est == PTAM_EST_TUKEY) return e2 > s2 ? 0.0 : 1.0 - (e2 / s2);
    if (est == PTAM_EST_CAUCHY) return sqrt(1.0 / (1.0 + e2 / s2));
    return sqrt(e2 < s2 ? 1.0 : sqrt(s2 / e2));
}
PTAM_HD double est_weight(int est, double e2, double s2) {
    if (est == PTAM_EST_TUKEY) {
        const double r = e2 > s2 ? 0.0 : 1.0 - (e2 / s2);
        return r * r;
    }
    if (est == PTAM_EST_CAUCHY) return 1.0 / (1.0 + e2 / s2);
    return e2 < s2 ? 1.0 : sqrt(s2 / e2);
}
PTAM_HD double est_objective(int est, double e2, double s2) {
    if (est == PTAM_EST_TUKEY) {
        if (e2 > s2) return 1.0;
        const double d = 1.0 - e2 / s2;
        return 1.0 - d * d * d;
    }
    if (est == PTAM_EST_CAUCHY) return log(1.0 + e2 / s2);
    if (e2 < s2) return 0.5 * e2;
    const double s = sqrt(s2), e = sqrt(e2);
    return s * (e - 0.5 * s);
}
// FindSigmaSquared from the median (sorted[n/2]) of n squared errors; size_t arithmetic in (2n-6)
PTAM_HD double est_sigma_sq_from_median(int est, double median_sq, unsigned long long n) {
    const unsigned long long den = n * 2ull - 6ull;   // wraps for n < 3 like the reference
    double sigma = 1.4826 * (1 + 5.0 / (double)den) * sqrt(median_sq);
    sigma = (est == PTAM_EST_HUBER ? 1.345 : 4.6851) * sigma;
    return sigma * sigma;
}

// ---- wave64 / block helpers (device only) -----------------------------------------------------
#ifdef __HIPCC__
// reciprocal: v_rcp_f64 + two Newton steps (<= 1 ulp), ~5 instructions instead of an IEEE division
__device__ __forceinline__ double rcp_nr(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
}
// DPP row shift right by SHR lanes inside each 16-lane row; lanes without a source keep `old`
template <int SHR>
__device__ __forceinline__ int dpp_row_shr_i32(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x110 + SHR, 0xf, 0xf, false);
}
// bound_ctrl form: lanes without a source read 0 (no `old` operand to materialise)
template <int SHR>
__device__ __forceinline__ int dpp_row_shr0_i32(int src) {
    return __builtin_amdgcn_update_dpp(src, src, 0x110 + SHR, 0xf, 0xf, true);
}
template <int SHR>
__device__ __forceinline__ double dpp_row_shr_f64(double src) {
    const long long b = __double_as_longlong(src);
    const int lo = dpp_row_shr0_i32<SHR>((int)(b & 0xffffffffll));
    const int hi = dpp_row_shr0_i32<SHR>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// DPP row_bcast15 (ctrl 0x142): lane 15 of row r -> every lane of row r+1 ; row_bcast31 (0x143):
// lane 31 -> every lane of rows 2,3.  row_mask selects the destination rows; others keep `old`.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_bcast_i32(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROWMASK, 0xf, false);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_bcast_f64(double src) {
    const long long b = __double_as_longlong(src);
    const int lo = dpp_bcast_i32<CTRL, ROWMASK>(0, (int)(b & 0xffffffffll));
    const int hi = dpp_bcast_i32<CTRL, ROWMASK>(0, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// full-row-mask form (no `old` to materialise): rows without a valid source hold garbage, the caller
// must discard them (row_bcast15: row 0 ; row_bcast31: rows 0,1)
template <int CTRL>
__device__ __forceinline__ int dpp_bcastx_i32(int src) {
    return __builtin_amdgcn_update_dpp(src, src, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_bcastx_f64(double src) {
    const long long b = __double_as_longlong(src);
    const int lo = dpp_bcastx_i32<CTRL>((int)(b & 0xffffffffll));
    const int hi = dpp_bcastx_i32<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Inclusive prefix sums over the 64 lanes, on the VALU (same DPP steps as the sums below; a __shfl_up ladder is six
// ds_bpermute round trips per 32 bits)
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    v += dpp_row_shr0_i32<1>(v);
    v += dpp_row_shr0_i32<2>(v);
    v += dpp_row_shr0_i32<4>(v);
    v += dpp_row_shr0_i32<8>(v);
    v += dpp_bcast_i32<0x142, 0xA>(0, v);
    v += dpp_bcast_i32<0x143, 0xC>(0, v);
    return v;
}
template <int SHR>
__device__ __forceinline__ long long dpp_row_shr_i64(long long b) {
    const int lo = dpp_row_shr0_i32<SHR>((int)(b & 0xffffffffll));
    const int hi = dpp_row_shr0_i32<SHR>((int)(b >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ long long dpp_bcast_i64(long long b) {
    const int lo = dpp_bcast_i32<CTRL, ROWMASK>(0, (int)(b & 0xffffffffll));
    const int hi = dpp_bcast_i32<CTRL, ROWMASK>(0, (int)(b >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long wave_incl_scan_i64(long long v) {
    v += dpp_row_shr_i64<1>(v);
    v += dpp_row_shr_i64<2>(v);
    v += dpp_row_shr_i64<4>(v);
    v += dpp_row_shr_i64<8>(v);
    v += dpp_bcast_i64<0x142, 0xA>(v);
    v += dpp_bcast_i64<0x143, 0xC>(v);
    return v;
}
// Sum over the 64 lanes, result in every lane.  On the VALU: inclusive scan inside the 16-lane rows (DPP row_shr
// 1, 2, 4, 8), lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15), lane 31 into rows 2 / 3 (row_bcast31) — the total is
// in lane 63 — and one v_readlane per 32 bits.  (Six __shfl_xor steps are six ds_bpermute round trips per 32 bits:
// ~1.4 us of a single-workgroup kernel's critical path for two fp64 sums.)  Fixed order: deterministic.
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += dpp_row_shr0_i32<1>(v);
    v += dpp_row_shr0_i32<2>(v);
    v += dpp_row_shr0_i32<4>(v);
    v += dpp_row_shr0_i32<8>(v);
    v += dpp_bcast_i32<0x142, 0xA>(0, v);
    v += dpp_bcast_i32<0x143, 0xC>(0, v);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_row_shr_f64<1>(v);
    v += dpp_row_shr_f64<2>(v);
    v += dpp_row_shr_f64<4>(v);
    v += dpp_row_shr_f64<8>(v);
    v += dpp_bcast_f64<0x142, 0xA>(v);
    v += dpp_bcast_f64<0x143, 0xC>(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#endif
