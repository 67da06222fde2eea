// patch.hip — PatchFinder::FindPatchCoarse (src/PatchFinder.cc:160-211) + ImageProcess::ZMSSDAtPoint
// (src/ImageProcess.cc:130-163) on gfx950.
//
// K3 zmssd_search_kernel: ONE WAVE PER PATCH.  An 8x8 patch is exactly one wave64: lane l holds
// template pixel T[l] in a register and scores pixel (l>>3, l&7) of a candidate window; the three
// sums (sum I, sum I^2, sum I*T) are reduced with wavefront shuffles.  Candidates are the FAST corners
// of rows [top, bottom] taken from the row LUT; the wave filters 64 corners at a time (x range +
// disc test) with a ballot and then walks the surviving bits in ascending order, which preserves the
// reference's "first strict minimum" tie-break (raster order).
#include "common.h"
#include <vector>

#include "keyframe.h"
#include "track_internal.h"
#include "patch_device.h"

// d_range (nullable): {first, end} query indices in device memory — the resident TrackMap chain decides on the device
// which slots a stage searches
__global__ void __launch_bounds__(256) zmssd_search_kernel(KfLevels L, int n, const ptam_patch_query* __restrict__ queries,
                                                           const uint8_t* __restrict__ templates,
                                                           ptam_patch_result* __restrict__ results, const int* __restrict__ d_range,
                                                           const ptam_template_result* __restrict__ tres) {
    const int lane = threadIdx.x & 63;
    int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d_range) {
        qi += d_range[0];
        n = min(n, d_range[1]);
    }
    if (qi >= n) return;
    const ptam_patch_query q = queries[qi];
    const int T = templates[(size_t)qi * 64 + lane];
    ptam_patch_result res;
    __shared__ __attribute__((aligned(16))) unsigned sw_win[4][SW_BYTES / 4];   // (a wave's own search region: no barrier)
    wave_find_patch_coarse(L, q, !(tres && tres[qi].bad), T, lane, res, sw_win[threadIdx.x >> 6]);
    if (lane == 0) results[qi] = res;
}

// ZMSSDAtPoint of one template at explicit points (one wave per point)
__global__ void __launch_bounds__(256) zmssd_points_kernel(KfLevels L, int lev, int n, const ptam_int2* __restrict__ pts,
                                                           const uint8_t* __restrict__ tmpl, int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int T = tmpl[lane];
    const int tsum = wave_sum_i32(T), tsumsq = wave_sum_i32(T * T);
    const ptam_int2 p = pts[i];
    const int ssd = wave_zmssd(L.im[lev], L.w[lev], L.h[lev], p.x, p.y, T, tsum, tsumsq, lane);
    if (lane == 0) out[i] = ssd;
}

// ------------------------------------------------------------------------------------------------
// PatchFinder::MakeSubPixTemplate + IterateSubPixToConvergence (src/PatchFinder.cc:219-318).
// One wave per patch, lane = template pixel (y = lane>>3, x = lane&7); the 36 interior lanes carry the
// inverse-compositional work.  Template gradients come from neighbouring lanes by shuffle; J^T J is a
// sum of multiples of 0.25 (exact in fp64, so order-free); the bilinear mix is done in fp32 with
// explicit round-to-nearest multiplies/adds in the reference's order (no FMA), so the interpolated
// pixel is bit-identical to the CPU's; the three J^T d sums are wave reductions.
// ------------------------------------------------------------------------------------------------
// Chain form (pq / pr given, queries == null): the query is the coarse search's own — level of pq[qi], position pr[qi].pos,
// only where pr[qi].found (src/Tracker.cc:897-905) — with chain_its iterations; d_range as in zmssd_search_kernel.
__global__ void __launch_bounds__(256) subpix_kernel(KfLevels L, int n, const ptam_subpix_query* __restrict__ queries,
                                                     const uint8_t* __restrict__ templates,
                                                     ptam_subpix_result* __restrict__ results, const int* __restrict__ d_range,
                                                     const ptam_patch_query* __restrict__ pq, const ptam_patch_result* __restrict__ pr,
                                                     int chain_its) {
    const int lane = threadIdx.x & 63;
    int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d_range) {
        qi += d_range[0];
        n = min(n, d_range[1]);
    }
    if (qi >= n) return;
    ptam_subpix_query q;
    if (pq) {
        const bool go = pq[qi].level >= 0 && pr[qi].found;
        q.level = go ? pq[qi].level : -1;
        q.max_its = chain_its;
        q.coarse_pos[0] = pr[qi].pos[0];
        q.coarse_pos[1] = pr[qi].pos[1];
    } else
        q = queries[qi];
    ptam_subpix_result res;
    const int T = (q.level >= 0 && q.level < PTAM_LEVELS) ? templates[(size_t)qi * 64 + lane] : 0;
    wave_subpix(L, q, T, lane, res);
    if (lane == 0) results[qi] = res;
}

// =================================================================================================
// MakeTemplateCoarseCont (src/PatchFinder.cc:98-127): CVD::transform of the source patch + template sums
// =================================================================================================
// (TemplateJob, the device-side form of ptam_template_query, lives in track_internal.h)

// One wave per template, lane = output pixel (i = lane / 8 row, j = lane % 8 column).  The source position is
// NOT evaluated in closed form: the reference walks p += across / += carriage_return pixel by pixel, and the
// lane replays that exact sequence of fp64 additions (<= 70 of them) so the sampled positions are bit-identical.
// Every product and sum goes through the nc_* primitives below: no FMA contraction.
__global__ void __launch_bounds__(256) make_templates_kernel(int n, const TemplateJob* __restrict__ jobs, uint8_t* __restrict__ tmpl,
                                                             ptam_template_result* __restrict__ res, const int* __restrict__ d_range) {
    int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (d_range) {
        q += d_range[0];
        n = min(n, d_range[1]);
    }
    if (q >= n) return;
    const TemplateJob jb = jobs[q];
    ptam_template_result r;
    const int v = wave_make_template(jb, lane, r);
    tmpl[(size_t)q * 64 + lane] = (uint8_t)v;
    if (lane == 0) res[q] = r;
}

// =================================================================================================
// AddPointEpipolar: in-plane corner table and the corner scan (src/MapMaker.cc:598-637)
// =================================================================================================
// ATANCamera::UnProject (src/ATANCamera.cc:125-140)
__device__ __forceinline__ void cam_unproject(const DevCam& c, double u, double v, double& x, double& y) {
    const double dx = (u - c.cx) * c.inv_fx, dy = (v - c.cy) * c.inv_fy;
    const double dr = sqrt(dx * dx + dy * dy);
    const double rr = (c.w == 0.0) ? dr : tan(dr * c.w) * c.one_over_two_tan;   // invrtrans include/ATANCamera.h:152-157
    const double f = dr > 0.01 ? rr / dr : 1.0;
    x = f * dx;
    y = f * dy;
}

// vv2Corners.push_back(imUnProj[ir(Level::LevelZeroPos(vIR[i], nLevel))])  :611-612
__global__ void __launch_bounds__(256) implane_corners_kernel(DevCam cam, KfLevels L, int lev, double2* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.ncorners[lev]) return;
    const ptam_int2 c = L.corners[lev][i];
    const int scale = 1 << lev;
    const int u = (int)((c.x + 0.5) * scale - 0.5), v = (int)((c.y + 0.5) * scale - 0.5);   // ir(): truncation
    double x, y;
    cam_unproject(cam, (double)u, (double)v, x, y);
    out[i] = make_double2(x, y);
}

// one wave per candidate: template = the 8x8 window of the source level (lane = pixel), then the target level's
// corners 64 at a time through the band / segment test, the survivors scored in corner order
__global__ void __launch_bounds__(256) epipolar_search_kernel(KfLevels S, KfLevels T, int lev, const double2* __restrict__ implane,
                                                              int n, const ptam_epipolar_query* __restrict__ queries,
                                                              ptam_epipolar_result* __restrict__ results) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= n) return;
    const ptam_epipolar_query q = queries[qi];
    ptam_epipolar_result res;
    res.best = -1;
    res.best_zmssd = PTAM_MAX_SSD + 1;
    res.n_scored = 0;
    res.template_bad = 0;
    const int sw = S.w[lev], sh = S.h[lev];
    // MakeTemplateCoarseNoWarp: in_image_with_border(irLevelPos, mnPatchSize / 2 + 1)
    if (!(q.level_x >= 5 && q.level_y >= 5 && q.level_x < sw - 5 && q.level_y < sh - 5)) {
        res.template_bad = 1;
        if (lane == 0) results[qi] = res;
        return;
    }
    const int Tp = S.im[lev][(size_t)(q.level_y - 4 + (lane >> 3)) * sw + (q.level_x - 4 + (lane & 7))];
    const int tsum = wave_sum_i32(Tp), tsumsq = wave_sum_i32(Tp * Tp);
    const unsigned T4 = wave_pack_template4(Tp, lane);
    const int tw = T.w[lev], th = T.h[lev];
    const uint8_t* im = T.im[lev];
    const ptam_int2* corners = T.corners[lev];
    const int nc = T.ncorners[lev];
    int best = PTAM_MAX_SSD + 1, bi = -1, nsc = 0;
    for (int base = 0; base < nc; base += 64) {
        const int idx = base + lane;
        bool pass = false;
        ptam_int2 c = {0, 0};
        if (idx < nc) {
            const double2 v = implane[idx];
            const double dd = q.norm_dist - (v.x * q.normal[0] + v.y * q.normal[1]);   // :623
            const double al = v.x * q.along[0] + v.y * q.along[1];
            pass = !(dd * dd > q.max_dist_sq) && !(al < q.min_len) && !(al > q.max_len);
            c = corners[idx];
        }
        unsigned long long m = __ballot(pass);
        while (m) {   // four candidates per pass (wave_zmssd4), judged in corner order
            int cx[4] = {0, 0, 0, 0}, cy[4] = {0, 0, 0, 0}, bb[4] = {0, 0, 0, 0}, nn = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (m) {   // (wave-uniform)
                    bb[k] = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    cx[k] = __builtin_amdgcn_readlane(c.x, bb[k]);
                    cy[k] = __builtin_amdgcn_readlane(c.y, bb[k]);
                    nn = k + 1;
                }
            const int ssd_l = wave_zmssd4(im, tw, th, cx, cy, nn, T4, tsum, tsumsq, lane);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < nn) {
                    const int ssd = __builtin_amdgcn_readlane(ssd_l, 16 * k + 15);
                    nsc++;
                    if (ssd < best) {
                        best = ssd;
                        bi = base + bb[k];
                    }
                }
        }
    }
    res.best = bi;
    res.best_zmssd = best;
    res.n_scored = nsc;
    if (lane == 0) results[qi] = res;
}

// ---- launch helpers of the resident TrackMap chain (track_internal.h) ------------------------------------------
int patch_launch_templates_dev(ptam_ctx* ctx, int n_cap, const TemplateJob* d_jobs, uint8_t* d_tmpl, ptam_template_result* d_res,
                               const int* d_range) {
    if (n_cap <= 0) return PTAM_OK;
    hipLaunchKernelGGL(make_templates_kernel, dim3((n_cap + 3) / 4), dim3(256), 0, ctx->stream, n_cap, d_jobs, d_tmpl, d_res, d_range);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}
int patch_launch_search_dev(ptam_ctx* ctx, const ptam_kf* kf, int n_cap, const ptam_patch_query* d_q, const uint8_t* d_tmpl,
                            ptam_patch_result* d_r, const int* d_range, const ptam_template_result* d_tres) {
    if (n_cap <= 0) return PTAM_OK;
    hipLaunchKernelGGL(zmssd_search_kernel, dim3((n_cap + 3) / 4), dim3(256), 0, ctx->stream, kf->L, n_cap, d_q, d_tmpl, d_r, d_range, d_tres);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}
int patch_launch_subpix_dev(ptam_ctx* ctx, const ptam_kf* kf, int n_cap, const ptam_patch_query* d_q, const ptam_patch_result* d_pr,
                            const uint8_t* d_tmpl, ptam_subpix_result* d_sr, const int* d_range, int max_its) {
    if (n_cap <= 0) return PTAM_OK;
    hipLaunchKernelGGL(subpix_kernel, dim3((n_cap + 3) / 4), dim3(256), 0, ctx->stream, kf->L, n_cap, (const ptam_subpix_query*)nullptr,
                       d_tmpl, d_sr, d_range, d_q, d_pr, max_its);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

extern "C" {

int ptam_subpix_batch(ptam_ctx* ctx, const ptam_kf* kf, int n, const ptam_subpix_query* queries, const uint8_t* templates,
                      ptam_subpix_result* results) {
    ARG_TRY(ctx && kf && n >= 0);
    if (n == 0) return PTAM_OK;
    ARG_TRY(queries && templates && results);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bq = (size_t)n * sizeof(ptam_subpix_query), bt = (size_t)n * 64, br = (size_t)n * sizeof(ptam_subpix_result);
    void* s;
    int rc = ctx_scratch(ctx, bq + bt + br, &s);
    if (rc) return rc;
    ptam_subpix_query* d_q = (ptam_subpix_query*)s;
    ptam_subpix_result* d_r = (ptam_subpix_result*)((char*)s + bq);
    uint8_t* d_t = (uint8_t*)s + bq + br;
    HIP_TRY(hipMemcpyAsync(d_q, queries, bq, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_t, templates, bt, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(subpix_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, kf->L, n, (const ptam_subpix_query*)d_q, (const uint8_t*)d_t, d_r,
                       (const int*)nullptr, (const ptam_patch_query*)nullptr, (const ptam_patch_result*)nullptr, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(results, d_r, br, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

int ptam_find_patch_coarse_batch_dev(ptam_ctx* ctx, const ptam_kf* kf, int n, const ptam_patch_query* d_q,
                                     const uint8_t* d_t, ptam_patch_result* d_r) {
    ARG_TRY(ctx && kf && n >= 0);
    if (n == 0) return PTAM_OK;
    ARG_TRY(d_q && d_t && d_r);
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(zmssd_search_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, kf->L, n, d_q, d_t, d_r, (const int*)nullptr,
                       (const ptam_template_result*)nullptr);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

int ptam_find_patch_coarse_batch(ptam_ctx* ctx, const ptam_kf* kf, int n, const ptam_patch_query* queries,
                                 const uint8_t* templates, ptam_patch_result* results) {
    ARG_TRY(ctx && kf && n >= 0);
    if (n == 0) return PTAM_OK;
    ARG_TRY(queries && templates && results);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bq = (size_t)n * sizeof(ptam_patch_query), bt = (size_t)n * 64, br = (size_t)n * sizeof(ptam_patch_result);
    void* s;
    int rc = ctx_scratch(ctx, bq + bt + br, &s);
    if (rc) return rc;
    ptam_patch_query* d_q = (ptam_patch_query*)s;
    ptam_patch_result* d_r = (ptam_patch_result*)((char*)s + bq);
    uint8_t* d_t = (uint8_t*)s + bq + br;
    HIP_TRY(hipMemcpyAsync(d_q, queries, bq, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_t, templates, bt, hipMemcpyHostToDevice, ctx->stream));
    rc = ptam_find_patch_coarse_batch_dev(ctx, kf, n, d_q, d_t, d_r);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(results, d_r, br, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

int ptam_zmssd_at_points(ptam_ctx* ctx, const ptam_kf* kf, int level, int n, const ptam_int2* points,
                         const uint8_t* tmpl64, int32_t* ssd_out) {
    ARG_TRY(ctx && kf && level >= 0 && level < PTAM_LEVELS && n >= 0);
    if (n == 0) return PTAM_OK;
    ARG_TRY(points && tmpl64 && ssd_out);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bp = (size_t)n * sizeof(ptam_int2), bo = (size_t)n * 4;
    void* s;
    int rc = ctx_scratch(ctx, bp + bo + 64, &s);
    if (rc) return rc;
    ptam_int2* d_p = (ptam_int2*)s;
    int* d_o = (int*)((char*)s + bp);
    uint8_t* d_t = (uint8_t*)s + bp + bo;
    HIP_TRY(hipMemcpyAsync(d_p, points, bp, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_t, tmpl64, 64, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(zmssd_points_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, kf->L, level, n, d_p, d_t, d_o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ssd_out, d_o, bo, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

int ptam_make_templates_batch(ptam_ctx* ctx, int n, const ptam_template_query* queries, uint8_t* templates_out,
                              ptam_template_result* results) {
    ARG_TRY(ctx && n >= 0);
    if (n == 0) return PTAM_OK;
    ARG_TRY(queries && templates_out && results);
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<TemplateJob> jobs((size_t)n);
    for (int i = 0; i < n; i++) {
        const ptam_template_query& q = queries[i];
        TemplateJob& j = jobs[(size_t)i];
        j.im = nullptr;
        j.w = j.h = 0;
        j.search_level = q.search_level;
        j.cx = q.center_x;
        j.cy = q.center_y;
        for (int k = 0; k < 4; k++) j.wi[k] = q.warp_inverse[k];
        if (q.search_level >= 0) {
            ARG_TRY(q.src_kf && q.src_level >= 0 && q.src_level < PTAM_LEVELS && q.search_level < PTAM_LEVELS);
            ARG_TRY(q.src_kf->device == ctx->device);
            j.im = q.src_kf->L.im[q.src_level];
            j.w = q.src_kf->L.w[q.src_level];
            j.h = q.src_kf->L.h[q.src_level];
        }
    }
    const size_t bj = (size_t)n * sizeof(TemplateJob), bt = (size_t)n * 64, br = (size_t)n * sizeof(ptam_template_result);
    void* s;
    int rc = ctx_scratch(ctx, bj + br + bt, &s);
    if (rc) return rc;
    TemplateJob* d_j = (TemplateJob*)s;
    ptam_template_result* d_r = (ptam_template_result*)((char*)s + bj);
    uint8_t* d_t = (uint8_t*)s + bj + br;
    HIP_TRY(hipMemcpyAsync(d_j, jobs.data(), bj, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(make_templates_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, n, (const TemplateJob*)d_j, d_t, d_r, (const int*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(templates_out, d_t, bt, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(results, d_r, br, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));   // (jobs[] is pageable: the H2D copy above has been staged by now)
    return PTAM_OK;
}

static int kf_build_implane(ptam_ctx* ctx, ptam_kf* kf, int level) {
    int rc = kf_fetch_counts(ctx, kf);
    if (rc) return rc;
    const int nc = kf->n_corners[level];
    if (kf->implane_valid[level]) return PTAM_OK;
    if (nc > kf->implane_cap[level]) {
        HIP_TRY(ptam_stream_wait(ctx->stream));
        if (kf->implane[level]) HIP_TRY(hipFree(kf->implane[level]));
        kf->implane[level] = nullptr;
        kf->implane_cap[level] = 0;
        const int cap = nc + nc / 4 + 64;
        HIP_TRY(hipMalloc((void**)&kf->implane[level], (size_t)cap * sizeof(double2)));
        kf->implane_cap[level] = cap;
    }
    if (nc > 0)
        hipLaunchKernelGGL(implane_corners_kernel, dim3((nc + 255) / 256), dim3(256), 0, ctx->stream, ctx->cam, kf->L, level,
                           kf->implane[level]);
    HIP_TRY(hipGetLastError());
    kf->implane_valid[level] = 1;
    return PTAM_OK;
}

int ptam_kf_implane_corners(ptam_ctx* ctx, ptam_kf* kf, int level, double* out_xy, int cap, int* n_out) {
    ARG_TRY(ctx && kf && level >= 0 && level < PTAM_LEVELS && cap >= 0);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = kf_build_implane(ctx, kf, level);
    if (rc) return rc;
    const int nc = kf->n_corners[level];
    if (n_out) *n_out = nc;
    if (out_xy) {
        ARG_TRY(cap >= nc);
        if (nc > 0) HIP_TRY(hipMemcpyAsync(out_xy, kf->implane[level], (size_t)nc * 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ptam_stream_wait(ctx->stream));
    }
    return PTAM_OK;
}

int ptam_epipolar_search_batch(ptam_ctx* ctx, const ptam_kf* src, ptam_kf* target, int level, int n,
                               const ptam_epipolar_query* queries, ptam_epipolar_result* results) {
    ARG_TRY(ctx && src && target && level >= 0 && level < PTAM_LEVELS && n >= 0);
    if (n == 0) return PTAM_OK;
    ARG_TRY(queries && results);
    ARG_TRY(src->device == ctx->device && target->device == ctx->device);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = kf_build_implane(ctx, target, level);
    if (rc) return rc;
    const size_t bq = (size_t)n * sizeof(ptam_epipolar_query), br = (size_t)n * sizeof(ptam_epipolar_result);
    void* s;
    rc = ctx_scratch(ctx, bq + br, &s);
    if (rc) return rc;
    ptam_epipolar_query* d_q = (ptam_epipolar_query*)s;
    ptam_epipolar_result* d_r = (ptam_epipolar_result*)((char*)s + bq);
    HIP_TRY(hipMemcpyAsync(d_q, queries, bq, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(epipolar_search_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, src->L, target->L, level,
                       (const double2*)target->implane[level], n, (const ptam_epipolar_query*)d_q, d_r);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(results, d_r, br, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

}   // extern "C"

// every kernel of this file resolved once, when a context is created: the first launch of a kernel otherwise pays for
// loading the code object / resolving the function — 10-28 ms in the middle of the first frame or the first adjustment
void patch_preload_kernels() {
    ptam_preload((const void*)zmssd_search_kernel);
    ptam_preload((const void*)zmssd_points_kernel);
    ptam_preload((const void*)subpix_kernel);
    ptam_preload((const void*)make_templates_kernel);
    ptam_preload((const void*)implane_corners_kernel);
    ptam_preload((const void*)epipolar_search_kernel);
}
