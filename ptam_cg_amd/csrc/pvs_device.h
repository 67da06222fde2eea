// pvs_device.h — device body of the PVS pass (pvs.hip wraps it as track_pvs_kernel; trackmap.hip runs it beside the pyramid
// of the new frame in one launch).
#pragma once
#include "common.h"
#include "track_internal.h"

// pv.use: the pose travels as a kernel argument (the resident TrackMap chain: the motion model's prediction needs no copy of
// its own) and block 0 also leaves it in pose_out for the kernels that follow
// mm2WarpInverse of CalcSearchLevelAndWarpMatrix (src/PatchFinder.cc:52-69) and its determinant: ONE copy for the PVS pass and the
// two ReFind kernels, uncontracted (see track_pvs_body)
__device__ __forceinline__ double pvs_warp_matrix(const double* T, double X, double Y, double Z, const double* D, const ptam_pvs_point& p, double* W) {
#pragma clang fp contract(off)
    const double iz = 1.0 / Z;
    double mr[3], md[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        mr[a] = T[a * 3] * p.pixel_right_w[0] + T[a * 3 + 1] * p.pixel_right_w[1] + T[a * 3 + 2] * p.pixel_right_w[2];
        md[a] = T[a * 3] * p.pixel_down_w[0] + T[a * 3 + 1] * p.pixel_down_w[1] + T[a * 3 + 2] * p.pixel_down_w[2];
    }
    const double ax = (mr[0] - X * mr[2] * iz) * iz, ay = (mr[1] - Y * mr[2] * iz) * iz;
    const double bx = (md[0] - X * md[2] * iz) * iz, by = (md[1] - Y * md[2] * iz) * iz;
    W[0] = D[0] * ax + D[1] * ay;
    W[2] = D[2] * ax + D[3] * ay;
    W[1] = D[0] * bx + D[1] * by;
    W[3] = D[2] * bx + D[3] * by;
    return W[0] * W[3] - W[1] * W[2];
}

__device__ __forceinline__ void track_pvs_body(const DevCam& cam, int n, const ptam_pvs_point* __restrict__ pts,
                                               const double* __restrict__ pose, ptam_pvs_result* __restrict__ out,
                                               int* __restrict__ counts, const PoseArg& pv, double* __restrict__ pose_out, int block,
                                               int* __restrict__ finder_bad = nullptr, int finder_stride = 0) {
    // No contraction of a * b + c * d in this function: which product the compiler fuses depends on the kernel the body is inlined
    // into, and the LAST BIT of the warp matrix decides grey levels of the warped template (CVD::transform truncates to a byte,
    // src/PatchFinder.cc:116) — the batch's PVS kernel and the single frame's gave 3-6 of ~1 000 templates one grey level apart
    // on maps of 2 000 points (tests/tools/batch_long_lists.py).  Plain products and sums are also what the reference's compiler emits.
#pragma clang fp contract(off)
    const int i = block * 256 + threadIdx.x;   // (a 256-thread workgroup)
    const int lane = threadIdx.x & 63;
    int level = -1;
    if (pv.use && pose_out && block == 0 && threadIdx.x < 12) pose_out[threadIdx.x] = pv.v[threadIdx.x];
    if (i < n) {
        double T[12];
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = pv.use ? pv.v[k] : pose[k];
        const ptam_pvs_point p = pts[i];
        ptam_pvs_result r;
        r.proj.image[0] = r.proj.image[1] = 0;
        r.proj.derivs[0] = r.proj.derivs[1] = r.proj.derivs[2] = r.proj.derivs[3] = 0;
        r.proj.in_image = 0;
        r.proj.pad_ = 0;
        r.warp_inverse[0] = r.warp_inverse[1] = r.warp_inverse[2] = r.warp_inverse[3] = 0;
        r.pad_ = 0;
        // TrackerData::Project include/Tracker.h:70-85
        se3_apply(T, p.world[0], p.world[1], p.world[2], r.proj.cam[0], r.proj.cam[1], r.proj.cam[2]);
        const double X = r.proj.cam[0], Y = r.proj.cam[1], Z = r.proj.cam[2];
        if (!(Z < 0.001)) {
            const double x = X / Z, y = Y / Z;
            if (!(x * x + y * y > cam.largest_radius * cam.largest_radius)) {
                double u, v, rr, f;
                cam_project(cam, x, y, u, v, rr, f);
                r.proj.image[0] = u;
                r.proj.image[1] = v;
                cam_derivs(cam, x, y, rr, f, r.proj.derivs);
                if (!(rr > cam.max_r) && !(u < 0 || v < 0 || u > cam.width || v > cam.height)) r.proj.in_image = 1;
            }
        }
        if (r.proj.in_image) {
            // CalcSearchLevelAndWarpMatrix src/PatchFinder.cc:52-84
            double* W = r.warp_inverse;   // mm2WarpInverse; .T()[0] / .T()[1] are its columns
            double det = pvs_warp_matrix(T, X, Y, Z, r.proj.derivs, p, W);
            int l = 0;
            while (det > 3 && l < PTAM_LEVELS - 1) {
                l++;
                det *= 0.25;
            }
            level = (det > 3 || det < 0.25) ? -1 : l;
            // the point's PatchFinder: a rejected warp sets mbTemplateBad (src/PatchFinder.cc:78-81), and the flag stays up until
            // the finder next re-makes its template (:98-127) — the resident tracker keeps it per point
            if (level < 0 && finder_bad) *(int*)((char*)finder_bad + (size_t)i * finder_stride) = 1;
        }
        r.level = level;
        out[i] = r;
    }
    if (counts) {
#pragma unroll
        for (int l = 0; l < PTAM_LEVELS; l++) {
            const unsigned long long m = __ballot(level == l);
            if (lane == 0 && m) atomicAdd(&counts[l], __popcll(m));
        }
    }
}

