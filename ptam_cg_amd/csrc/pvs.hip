// pvs.hip — Tracker::TrackMap's potentially-visible-set loop (src/Tracker.cc:453-478): project every
// map point, fetch the camera derivatives, and run PatchFinder::CalcSearchLevelAndWarpMatrix
// (src/PatchFinder.cc:52-84) — the search pyramid level and the 2x2 warp of a source-image pixel
// step into the current view.  One thread per map point; the per-level PVS sizes (avPVS[l].size())
// are counted with one atomic per wave and level.  
#include "common.h"
#include "track_internal.h"

// pv.use: the pose travels as a kernel argument (the resident TrackMap chain: the motion model's prediction needs no copy of
// its own) and block 0 also leaves it in pose_out for the kernels that follow
__global__ void __launch_bounds__(256) track_pvs_kernel(DevCam cam, int n, const ptam_pvs_point* __restrict__ pts,
                                                        const double* __restrict__ pose, ptam_pvs_result* __restrict__ out,
                                                        int* __restrict__ counts, PoseArg pv, double* __restrict__ pose_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int level = -1;
    if (pv.use && pose_out && blockIdx.x == 0 && threadIdx.x < 12) pose_out[threadIdx.x] = pv.v[threadIdx.x];
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
            const double* D = r.proj.derivs;
            const double iz = 1.0 / Z;
            double mr[3], md[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                mr[a] = T[a * 3] * p.pixel_right_w[0] + T[a * 3 + 1] * p.pixel_right_w[1] + T[a * 3 + 2] * p.pixel_right_w[2];
                md[a] = T[a * 3] * p.pixel_down_w[0] + T[a * 3 + 1] * p.pixel_down_w[1] + T[a * 3 + 2] * p.pixel_down_w[2];
            }
            const double ax = (mr[0] - X * mr[2] * iz) * iz, ay = (mr[1] - Y * mr[2] * iz) * iz;
            const double bx = (md[0] - X * md[2] * iz) * iz, by = (md[1] - Y * md[2] * iz) * iz;
            double* W = r.warp_inverse;   // mm2WarpInverse; .T()[0] / .T()[1] are its columns
            W[0] = D[0] * ax + D[1] * ay;
            W[2] = D[2] * ax + D[3] * ay;
            W[1] = D[0] * bx + D[1] * by;
            W[3] = D[2] * bx + D[3] * by;
            double det = W[0] * W[3] - W[1] * W[2];
            int l = 0;
            while (det > 3 && l < PTAM_LEVELS - 1) {
                l++;
                det *= 0.25;
            }
            level = (det > 3 || det < 0.25) ? -1 : l;
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

int pvs_launch_dev(ptam_ctx* ctx, int n, const ptam_pvs_point* d_pts, double* d_pose, const double* host_pose, ptam_pvs_result* d_out) {
    if (n <= 0 && !host_pose) return PTAM_OK;   // (an empty map still has to leave the pose for the kernels that follow)
    PoseArg pv{};
    if (host_pose) {
        std::memcpy(pv.v, host_pose, 96);
        pv.use = 1;
    }
    hipLaunchKernelGGL(track_pvs_kernel, dim3(std::max(1, (n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->cam, std::max(n, 0), d_pts,
                       (const double*)d_pose, d_out, (int*)nullptr, pv, d_pose);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

extern "C" int ptam_track_pvs(ptam_ctx* ctx, int n, const ptam_pvs_point* points, const double pose[12],
                              ptam_pvs_result* results, int32_t counts[4]) {
    ARG_TRY(ctx && n >= 0 && pose);
    if (counts) counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if (n == 0) return PTAM_OK;
    ARG_TRY(points && results);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bp = (size_t)n * sizeof(ptam_pvs_point), br = (size_t)n * sizeof(ptam_pvs_result);
    void* s;
    int rc = ctx_scratch(ctx, bp + br + 96 + 64, &s);
    if (rc) return rc;
    ptam_pvs_point* d_p = (ptam_pvs_point*)s;
    ptam_pvs_result* d_r = (ptam_pvs_result*)((char*)s + bp);
    double* d_pose = (double*)((char*)s + bp + br);
    int* d_c = (int*)((char*)s + bp + br + 96);
    HIP_TRY(hipMemcpyAsync(d_p, points, bp, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_pose, pose, 96, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_c, 0, 16, ctx->stream));
    hipLaunchKernelGGL(track_pvs_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->cam, n, d_p, (const double*)d_pose, d_r, d_c, PoseArg{},
                       (double*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(results, d_r, br, hipMemcpyDeviceToHost, ctx->stream));
    if (counts) HIP_TRY(hipMemcpyAsync(counts, d_c, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

// every kernel of this file resolved once, when a context is created: the first launch of a kernel otherwise pays for
// loading the code object / resolving the function — 10-28 ms in the middle of the first frame or the first adjustment
void pvs_preload_kernels() {
    ptam_preload((const void*)track_pvs_kernel);
}
