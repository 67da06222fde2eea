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
#include "keyframe.h"

// integer zero-mean SSD; the /64 is a C division truncating toward zero (numerator <= 0)
__device__ __forceinline__ int zmssd_finish(int SA, int SB, int isumsq, int tsumsq, int cross) {
    return ((2 * SA * SB - SA * SA - SB * SB) / 64 + isumsq + tsumsq - 2 * cross);
}

// score the window centred on (cx,cy) of one level against the wave's template (T = this lane's pixel)
__device__ __forceinline__ int wave_zmssd(const uint8_t* __restrict__ im, int w, int h, int cx, int cy, int T,
                                          int tsum, int tsumsq, int lane) {
    if (!(cx >= 4 && cy >= 4 && cx < w - 4 && cy < h - 4)) return PTAM_MAX_SSD + 1;
    const int I = im[(size_t)(cy - 4 + (lane >> 3)) * w + (cx - 4 + (lane & 7))];
    const int isum = wave_sum_i32(I);
    const int isumsq = wave_sum_i32(I * I);
    const int cross = wave_sum_i32(I * T);
    return zmssd_finish(tsum, isum, isumsq, tsumsq, cross);
}

__global__ void __launch_bounds__(256) zmssd_search_kernel(KfLevels L, int n, const ptam_patch_query* __restrict__ queries,
                                                           const uint8_t* __restrict__ templates,
                                                           ptam_patch_result* __restrict__ results) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= n) return;
    const ptam_patch_query q = queries[qi];
    ptam_patch_result res;
    res.found = 0;
    res.best_ssd = PTAM_MAX_SSD + 1;
    res.best_x = res.best_y = -1;
    res.n_scored = 0;
    res.pad_ = 0;
    res.pos[0] = res.pos[1] = 0;
    bool search = q.level >= 0 && q.level < PTAM_LEVELS;
    int w = 0, h = 0, px = 0, py = 0, nLeft = 0, nRight = 0, i0 = 0, i1 = 0;
    unsigned nRange = 0;
    const uint8_t* im = nullptr;
    const ptam_int2* corners = nullptr;
    if (search) {
        const int lev = q.level;
        w = L.w[lev];
        h = L.h[lev];
        im = L.im[lev];
        corners = L.corners[lev];
        const int scale = 1 << lev;
        px = q.x / scale;   // ImageRef / int: C division
        py = q.y / scale;
        nRange = (q.range + scale - 1) / scale;
        int nTop = (int)((unsigned)py - nRange);
        const int nBottomPlusOne = (int)((unsigned)py + nRange + 1u);
        nLeft = (int)((unsigned)px - nRange);
        nRight = (int)((unsigned)px + nRange);
        if (nTop < 0) nTop = 0;
        if (nTop >= h || nBottomPlusOne <= 0)
            search = false;
        else {
            i0 = L.rowlut[lev][nTop];
            i1 = nBottomPlusOne >= h ? L.ncorners[lev] : L.rowlut[lev][nBottomPlusOne];
        }
    }
    if (search) {
        const int T = templates[(size_t)qi * 64 + lane];
        const int tsum = wave_sum_i32(T), tsumsq = wave_sum_i32(T * T);
        int best = PTAM_MAX_SSD + 1, bx = -1, by = -1, nsc = 0;
        for (int base = i0; base < i1; base += 64) {
            const int idx = base + lane;
            ptam_int2 c = {0, 0};
            bool pass = false;
            if (idx < i1) {
                c = corners[idx];
                const int dx = px - c.x, dy = py - c.y;
                pass = !(c.x < nLeft || c.x > nRight) && !((unsigned)(dx * dx + dy * dy) > nRange * nRange);
            }
            unsigned long long m = __ballot(pass);
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int cx = __shfl(c.x, b, 64), cy = __shfl(c.y, b, 64);
                const int ssd = wave_zmssd(im, w, h, cx, cy, T, tsum, tsumsq, lane);
                nsc++;
                if (ssd < best) {
                    best = ssd;
                    bx = cx;
                    by = cy;
                }
            }
        }
        res.best_ssd = best;
        res.best_x = bx;
        res.best_y = by;
        res.n_scored = nsc;
        if (best < PTAM_MAX_SSD) {
            const int scale = 1 << q.level;
            res.found = 1;
            res.pos[0] = (bx + 0.5) * scale - 0.5;   // Level::LevelZeroPos include/KeyFrame.h:91-94
            res.pos[1] = (by + 0.5) * scale - 0.5;
        }
    }
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

extern "C" {

int ptam_find_patch_coarse_batch_dev(ptam_ctx* ctx, const ptam_kf* kf, int n, const ptam_patch_query* d_q,
                                     const uint8_t* d_t, ptam_patch_result* d_r) {
    ARG_TRY(ctx && kf && n >= 0);
    if (n == 0) return PTAM_OK;
    ARG_TRY(d_q && d_t && d_r);
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(zmssd_search_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, kf->L, n, d_q, d_t, d_r);
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
    HIP_TRY(hipStreamSynchronize(ctx->stream));
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
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PTAM_OK;
}

}   // extern "C"
