// ctx.hip — context, error reporting, device-memory helpers, projection batch (a8/a12/a13).
#include <chrono>
#include <cstdarg>

#include "common.h"

static thread_local char g_err[512] = "";

void ptam_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int ctx_scratch(ptam_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->d_scratch_cap) {
        HIP_TRY(ptam_stream_wait(ctx->stream));
        if (ctx->d_scratch) HIP_TRY(hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->d_scratch_cap = 0;
        size_t cap = bytes + bytes / 2 + 4096;
        HIP_TRY(hipMalloc(&ctx->d_scratch, cap));
        ctx->d_scratch_cap = cap;
    }
    *out = ctx->d_scratch;
    return PTAM_OK;
}

void ptam_preload(const void* kernel) {
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, kernel) != hipSuccess) (void)hipGetLastError();
}

hipError_t ptam_stream_wait(hipStream_t stream) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(stream);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    (void)hipGetLastError();   // (hipErrorNotReady is sticky in the last-error slot)
    return hipStreamSynchronize(stream);
}

int ctx_cache_take(ptam_ctx::Cached* c, int slots, size_t bytes, void** out, size_t* cap) {
    int best = -1;
    for (int i = 0; i < slots; i++)
        if (c[i].p && c[i].bytes >= bytes && (best < 0 || c[i].bytes < c[best].bytes)) best = i;
    *out = nullptr;
    *cap = 0;
    if (best >= 0) {
        *out = c[best].p;
        *cap = c[best].bytes;
        c[best].p = nullptr;
        c[best].bytes = 0;
    }
    return best >= 0;
}
void* ctx_cache_give(ptam_ctx::Cached* c, int slots, void* p, size_t bytes) {
    if (!p) return nullptr;
    for (int i = 0; i < slots; i++)
        if (!c[i].p) {
            c[i].p = p;
            c[i].bytes = bytes;
            return nullptr;
        }
    int small = 0;   // every slot taken: keep the largest
    for (int i = 1; i < slots; i++)
        if (c[i].bytes < c[small].bytes) small = i;
    if (bytes <= c[small].bytes) return p;
    void* drop = c[small].p;
    c[small].p = p;
    c[small].bytes = bytes;
    return drop;
}

int ctx_pinned(ptam_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->h_pinned_cap) {
        HIP_TRY(ptam_stream_wait(ctx->stream));
        if (ctx->h_pinned) HIP_TRY(hipHostFree(ctx->h_pinned));
        ctx->h_pinned = nullptr;
        ctx->h_pinned_cap = 0;
        size_t cap = bytes + bytes / 2 + 4096;
        HIP_TRY(hipHostMalloc(&ctx->h_pinned, cap, hipHostMallocMapped | hipHostMallocCoherent));
        ctx->h_pinned_cap = cap;
        std::memset(ctx->h_pinned, 0, cap);
        HIP_TRY(hipHostGetDevicePointer(&ctx->d_pinned, ctx->h_pinned, 0));
    }
    *out = ctx->h_pinned;
    return PTAM_OK;
}

// RefreshParams src/ATANCamera.cc:27-66 (only the members the hot path reads)
static DevCam make_devcam(const ptam_cam_params& p) {
    DevCam c;
    c.width = p.width;
    c.height = p.height;
    c.fx = c.width * p.fx;
    c.fy = c.height * p.fy;
    c.cx = c.width * p.cx - 0.5;
    c.cy = c.height * p.cy - 0.5;
    c.w = p.w;
    double one_over_two_tan;
    if (c.w != 0.0) {
        c.two_tan = 2.0 * std::tan(c.w / 2.0);
        one_over_two_tan = 1.0 / c.two_tan;
        c.w_inv = 1.0 / c.w;
        c.dist_enabled = 1.0;
    } else {
        c.w_inv = 0.0;
        c.two_tan = 0.0;
        one_over_two_tan = 0.0;
        c.dist_enabled = 0.0;
    }
    const double vx = std::fmax(p.cx, 1.0 - p.cx) / p.fx;
    const double vy = std::fmax(p.cy, 1.0 - p.cy) / p.fy;
    const double r = std::sqrt(vx * vx + vy * vy);
    c.largest_radius = (c.w == 0.0) ? r : std::tan(r * c.w) * one_over_two_tan;   // invrtrans
    c.max_r = 1.5 * c.largest_radius;
    c.inv_fx = 1.0 / c.fx;   // :38-39
    c.inv_fy = 1.0 / c.fy;
    c.one_over_two_tan = one_over_two_tan;
    // mdOnePixelDist :69-75 — UnProject at the image centre and one pixel down-right of it
    auto unproject = [&](double u, double v, double out[2]) {
        const double dx = (u - c.cx) * c.inv_fx, dy = (v - c.cy) * c.inv_fy;
        const double dr = std::sqrt(dx * dx + dy * dy);
        const double rr = (c.w == 0.0) ? dr : std::tan(dr * c.w) * one_over_two_tan;
        const double f = dr > 0.01 ? rr / dr : 1.0;
        out[0] = f * dx;
        out[1] = f * dy;
    };
    double a[2], b[2];
    unproject(p.width / 2.0, p.height / 2.0, a);           // mvImageSize is a Vector<2> of doubles
    unproject(p.width / 2.0 + 1, p.height / 2.0 + 1, b);
    const double ddx = a[0] - b[0], ddy = a[1] - b[1];
    c.one_pixel_dist = std::sqrt(ddx * ddx + ddy * ddy) / std::sqrt(2.0);
    return c;
}

// TrackerData::Project + GetProjectionDerivs (include/Tracker.h:70-94)
__global__ void project_points_kernel(DevCam cam, int n, const double* __restrict__ world,
                                      const double* __restrict__ pose, ptam_projection* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = pose[k];
    ptam_projection p;
    p.image[0] = p.image[1] = 0;
    p.derivs[0] = p.derivs[1] = p.derivs[2] = p.derivs[3] = 0;
    p.in_image = 0;
    p.pad_ = 0;
    se3_apply(T, world[3 * i], world[3 * i + 1], world[3 * i + 2], p.cam[0], p.cam[1], p.cam[2]);
    if (!(p.cam[2] < 0.001)) {
        const double x = p.cam[0] / p.cam[2], y = p.cam[1] / p.cam[2];
        if (!(x * x + y * y > cam.largest_radius * cam.largest_radius)) {
            double u, v, r, f;
            cam_project(cam, x, y, u, v, r, f);
            p.image[0] = u;
            p.image[1] = v;
            cam_derivs(cam, x, y, r, f, p.derivs);
            const bool invalid = r > cam.max_r;
            if (!invalid && !(u < 0 || v < 0 || u > cam.width || v > cam.height)) p.in_image = 1;
        }
    }
    out[i] = p;
}

// TrackerData::Project on existing state (include/Tracker.h:70-85): v3Cam always, v2Image once the camera model is reached,
// m2CamDerivs untouched (ProjectAndDerivs only refreshes them for found points, :89-94)
__global__ void reproject_points_kernel(DevCam cam, int n, const double* __restrict__ world, const double* __restrict__ pose,
                                        ptam_projection* __restrict__ io) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = pose[k];
    ptam_projection p = io[i];
    p.in_image = 0;
    se3_apply(T, world[3 * i], world[3 * i + 1], world[3 * i + 2], p.cam[0], p.cam[1], p.cam[2]);
    if (!(p.cam[2] < 0.001)) {
        const double x = p.cam[0] / p.cam[2], y = p.cam[1] / p.cam[2];
        if (!(x * x + y * y > cam.largest_radius * cam.largest_radius)) {
            double u, v, r, f;
            cam_project(cam, x, y, u, v, r, f);
            p.image[0] = u;
            p.image[1] = v;
            if (!(r > cam.max_r) && !(u < 0 || v < 0 || u > cam.width || v > cam.height)) p.in_image = 1;
        }
    }
    io[i] = p;
}

extern "C" {

const char* ptam_last_error(void) { return g_err; }

int ptam_device_count(int* n) {
    ARG_TRY(n);
    HIP_TRY(hipGetDeviceCount(n));
    return PTAM_OK;
}

int ptam_ctx_create(const ptam_cam_params* cam, int device, ptam_ctx** out) {
    ARG_TRY(cam && out);
    ARG_TRY(cam->width > 0 && cam->height > 0 && cam->fx != 0 && cam->fy != 0);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0 || device < 0 || device >= ndev) {
        ptam_set_error("no HIP device %d (count %d): libptam_hip has no CPU fallback", device, ndev);
        return PTAM_E_HIP;
    }
    HIP_TRY(hipSetDevice(device));
    ptam_ctx* c = new ptam_ctx();
    std::memset(c, 0, sizeof *c);
    c->device = device;
    c->params = *cam;
    c->cam = make_devcam(*cam);
    c->halfsample = PTAM_HALFSAMPLE_R;
    {
        hipDeviceProp_t prop;
        c->n_cu = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        ptam_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete c;
        return PTAM_E_HIP;
    }
    // (the second queue of a bundle's rejected trials, bundle.hip: created here — creating a queue while kernels are queued elsewhere
    //  stalls them for milliseconds)
    if (hipStreamCreateWithFlags(&c->stream_alt, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        c->stream_alt = nullptr;
    }
    ptam_preload((const void*)project_points_kernel);
    ptam_preload((const void*)reproject_points_kernel);
    trackmap_preload_kernels();
    ba_preload_kernels();
    solve_preload_kernels();
    pose_preload_kernels();
    patch_preload_kernels();
    kf_preload_kernels();
    pvs_preload_kernels();
    *out = c;
    return PTAM_OK;
}

int ptam_ctx_destroy(ptam_ctx* ctx) {
    if (!ctx) return PTAM_OK;
    hipSetDevice(ctx->device);
    ptam_stream_wait(ctx->stream);
    if (ctx->d_scratch) hipFree(ctx->d_scratch);
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    for (int i = 0; i < CTX_NCACHE(ctx->dev_cache); i++)
        if (ctx->dev_cache[i].p) hipFree(ctx->dev_cache[i].p);
    for (int i = 0; i < CTX_NCACHE(ctx->host_cache); i++)
        if (ctx->host_cache[i].p) hipHostFree(ctx->host_cache[i].p);
    for (int i = 0; i < CTX_NCACHE(ctx->pin_cache); i++)
        if (ctx->pin_cache[i].p) hipHostFree(ctx->pin_cache[i].p);
    if (ctx->d_smap) hipFree(ctx->d_smap);
    if (ctx->stream_alt) hipStreamDestroy(ctx->stream_alt);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return PTAM_OK;
}

int ptam_ctx_set_halfsample(ptam_ctx* ctx, int variant) {
    ARG_TRY(ctx && (variant == PTAM_HALFSAMPLE_R || variant == PTAM_HALFSAMPLE_T));
    ctx->halfsample = variant;
    return PTAM_OK;
}

int ptam_ctx_sync(ptam_ctx* ctx) {
    ARG_TRY(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

void* ptam_ctx_stream(ptam_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ptam_ctx_camera_constants(ptam_ctx* ctx, double out[8]) {
    ARG_TRY(ctx && out);
    out[0] = ctx->cam.fx;
    out[1] = ctx->cam.fy;
    out[2] = ctx->cam.cx;
    out[3] = ctx->cam.cy;
    out[4] = ctx->cam.two_tan;
    out[5] = ctx->cam.w_inv;
    out[6] = ctx->cam.largest_radius;
    out[7] = ctx->cam.max_r;
    return PTAM_OK;
}

int ptam_ctx_cache_hazards(ptam_ctx* ctx, long long* out) {
    ARG_TRY(ctx && out);
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long a = 0, b = 0;
    int rc = pose_hazards_read_pose(ctx->stream, &a);
    if (!rc) rc = pose_hazards_read_trackmap(ctx->stream, &b);
    if (rc) return rc;
    HIP_TRY(ptam_stream_wait(ctx->stream));
    *out = (long long)(a + b);
    return PTAM_OK;
}

int ptam_ctx_one_pixel_dist(ptam_ctx* ctx, double* out) {
    ARG_TRY(ctx && out);
    *out = ctx->cam.one_pixel_dist;
    return PTAM_OK;
}

int ptam_dev_alloc(ptam_ctx* ctx, size_t bytes, void** dptr) {
    ARG_TRY(ctx && dptr);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return PTAM_OK;
}
int ptam_dev_free(ptam_ctx* ctx, void* dptr) {
    ARG_TRY(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    HIP_TRY(hipFree(dptr));
    return PTAM_OK;
}
int ptam_dev_upload(ptam_ctx* ctx, void* dptr, const void* host, size_t bytes) {
    ARG_TRY(ctx && dptr && host);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}
int ptam_dev_download(ptam_ctx* ctx, void* host, const void* dptr, size_t bytes) {
    ARG_TRY(ctx && dptr && host);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

int ptam_project_points(ptam_ctx* ctx, int n, const double* world_xyz, const double pose[12],
                        ptam_projection* out) {
    ARG_TRY(ctx && n >= 0 && pose && (n == 0 || (world_xyz && out)));
    if (n == 0) return PTAM_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t wb = (size_t)n * 24, ob = (size_t)n * sizeof(ptam_projection);
    void* scratch;
    int rc = ctx_scratch(ctx, wb + 128 + ob, &scratch);
    if (rc) return rc;
    double* d_world = (double*)scratch;
    double* d_pose = (double*)((char*)scratch + wb);
    ptam_projection* d_out = (ptam_projection*)((char*)scratch + wb + 128);
    HIP_TRY(hipMemcpyAsync(d_world, world_xyz, wb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_pose, pose, 96, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(project_points_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->cam, n,
                       d_world, d_pose, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

int ptam_reproject_points(ptam_ctx* ctx, int n, const double* world_xyz, const double pose[12], ptam_projection* inout) {
    ARG_TRY(ctx && n >= 0 && pose && (n == 0 || (world_xyz && inout)));
    if (n == 0) return PTAM_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t wb = (size_t)n * 24, ob = (size_t)n * sizeof(ptam_projection);
    void* scratch;
    int rc = ctx_scratch(ctx, wb + 128 + ob, &scratch);
    if (rc) return rc;
    double* d_world = (double*)scratch;
    double* d_pose = (double*)((char*)scratch + wb);
    ptam_projection* d_io = (ptam_projection*)((char*)scratch + wb + 128);
    HIP_TRY(hipMemcpyAsync(d_world, world_xyz, wb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_pose, pose, 96, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_io, inout, ob, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(reproject_points_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->cam, n, d_world, d_pose, d_io);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(inout, d_io, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

}   // extern "C"
