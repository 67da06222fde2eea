// pvs.hip — Tracker::TrackMap's potentially-visible-set loop (src/Tracker.cc:453-478): project every
// map point, fetch the camera derivatives, and run PatchFinder::CalcSearchLevelAndWarpMatrix
// (src/PatchFinder.cc:52-84) — the search pyramid level and the 2x2 warp of a source-image pixel
// step into the current view.  One thread per map point; the per-level PVS sizes (avPVS[l].size())
// are counted with one atomic per wave and level.  
#include "common.h"
#include "track_internal.h"
#include "pvs_device.h"

__global__ void __launch_bounds__(256) track_pvs_kernel(DevCam cam, int n, const ptam_pvs_point* __restrict__ pts,
                                                        const double* __restrict__ pose, ptam_pvs_result* __restrict__ out,
                                                        int* __restrict__ counts, PoseArg pv, double* __restrict__ pose_out,
                                                        int* __restrict__ finder_bad, int finder_stride) {
    track_pvs_body(cam, n, pts, pose, out, counts, pv, pose_out, blockIdx.x, finder_bad, finder_stride);
}

int pvs_launch_dev(ptam_ctx* ctx, int n, const ptam_pvs_point* d_pts, double* d_pose, const double* host_pose, ptam_pvs_result* d_out,
                   int* d_finder_bad, int finder_stride) {
    if (n <= 0 && !host_pose) return PTAM_OK;   // (an empty map still has to leave the pose for the kernels that follow)
    PoseArg pv{};
    if (host_pose) {
        std::memcpy(pv.v, host_pose, 96);
        pv.use = 1;
    }
    hipLaunchKernelGGL(track_pvs_kernel, dim3(std::max(1, (n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->cam, std::max(n, 0), d_pts,
                       (const double*)d_pose, d_out, (int*)nullptr, pv, d_pose, d_finder_bad, finder_stride);
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
                       (double*)nullptr, (int*)nullptr, 0);
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
