// keyframe.hip — KeyFrame::MakeKeyFrame_Lite (src/KeyFrame.cc:18-54) on gfx950:
//   K1  pyramid_kernel       3x halfSample in one launch (register cascade over an 8x8 L0 block)
//   K2a fast_detect_kernel   FAST-10 on all 4 levels in one launch; LDS-staged tiles with halo;
//                            one wave = one 64-pixel row segment, corner bit-mask by wave ballot
//   K2b fast_compact_kernel  raster-ordered compaction: popcount prefix over the (row, tile) masks
//                            -> corner list + row LUT (the LUT IS the exclusive row prefix)
#include "common.h"
#include "keyframe.h"
#include "keyframe_device.h"
#include "track_internal.h"

// ------------------------------------------------------------------------------------------------
// K1: halfSample cascade (body: keyframe_device.h).  Thread = one 8x8 block of L0.
// ------------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ void __launch_bounds__(256) pyramid_kernel(PyrArgs a) {
    pyramid_body<VARIANT>(a, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y);
}

// ------------------------------------------------------------------------------------------------
// K2a: FAST-10 segment test (libCVD fast_corner_detect_10 semantics).
// Block = 256 threads = 4 waves = a 64 x 4 pixel tile; LDS tile carries a 3-pixel halo.
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool has_run10(unsigned m16) {
    const unsigned d = m16 | (m16 << 16);
    const unsigned a = d & (d >> 1);
    const unsigned b = a & (a >> 2);
    const unsigned c = b & (b >> 4);
    return (c & (a >> 8) & 0xffffu) != 0;
}

__device__ __forceinline__ void fast_detect_body(const KfLevels& L, int bx) {   // a 256-thread workgroup
    __shared__ uint8_t tile[(FAST_TH + 6) * FAST_LW];
    // which level does this block belong to?
    int lev = 0;
#pragma unroll
    for (int l = 1; l < PTAM_LEVELS; l++)
        if (bx >= L.block_begin[l]) lev = l;
    const int b = bx - L.block_begin[lev];
    const int w = L.w[lev], h = L.h[lev], ntx = L.ntx[lev];
    const int tx = b % ntx, ty = b / ntx;
    const int x0 = tx * FAST_TW, y0 = ty * FAST_TH;
    const uint8_t* __restrict__ im = L.im[lev];
    // stage (TH+6) x (TW+6) pixels, origin (x0-3, y0-3); out-of-image pixels read as 0 (never used
    // by an in-range centre)
    for (int i = threadIdx.x; i < (FAST_TH + 6) * (FAST_TW + 6); i += 256) {
        const int r = i / (FAST_TW + 6), c = i - r * (FAST_TW + 6);
        const int x = x0 - 3 + c, y = y0 - 3 + r;
        uint8_t v = 0;
        if (x >= 0 && x < w && y >= 0 && y < h) v = im[(size_t)y * w + x];
        tile[r * FAST_LW + c] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = x0 + lx, y = y0 + ly;
    bool corner = false;
    if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
        const uint8_t* c = &tile[(ly + 3) * FAST_LW + lx + 3];
        const int v = *c, hi = v + L.thr[lev], lo = v - L.thr[lev];
        // ring order (dx,dy): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
        const int ring[16] = {c[3 * FAST_LW],      c[3 * FAST_LW + 1],  c[2 * FAST_LW + 2],  c[FAST_LW + 3],
                              c[3],                c[-FAST_LW + 3],     c[-2 * FAST_LW + 2], c[-3 * FAST_LW + 1],
                              c[-3 * FAST_LW],     c[-3 * FAST_LW - 1], c[-2 * FAST_LW - 2], c[-FAST_LW - 3],
                              c[-3],               c[FAST_LW - 3],      c[2 * FAST_LW - 2],  c[3 * FAST_LW - 1]};
        unsigned br = 0, dk = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            br |= (unsigned)(ring[i] > hi) << i;
            dk |= (unsigned)(ring[i] < lo) << i;
        }
        corner = has_run10(br) || has_run10(dk);
    }
    const unsigned long long m = __ballot(corner);
    if (lx == 0 && y < h) L.mask[lev][(size_t)y * ntx + tx] = m;
}

// The same test on a TALL tile, for launches that have workgroups to spare (a batch of frames): 64 x TH pixels per workgroup,
// wave w takes rows w, w + 4, ...; the halo costs (TH + 6) / TH instead of 10 / 4, and the tile is staged as aligned 32-bit
// words from column x0 - 4 (18 words per row) where the image width allows it.  Workgroups are numbered level by level,
// ntx * ceil(h / TH) each.
template <int TH>
__device__ __forceinline__ void fast_detect_tall_body(const KfLevels& L, int bx) {   // a 256-thread workgroup
    __shared__ __attribute__((aligned(4))) uint8_t tile[(TH + 6) * FAST_LW];
    static_assert(FAST_LW == FAST_TW + 8 && TH % 4 == 0, "tile pitch = 18 words");
    int lev = 0, b = bx;
    for (; lev < PTAM_LEVELS - 1; lev++) {
        const int nb = L.ntx[lev] * ((L.h[lev] + TH - 1) / TH);
        if (b < nb) break;
        b -= nb;
    }
    const int w = L.w[lev], h = L.h[lev], ntx = L.ntx[lev];
    if (b >= ntx * ((h + TH - 1) / TH)) return;
    const int tx = b % ntx, ty = b / ntx;
    const int x0 = tx * FAST_TW, y0 = ty * TH;
    const uint8_t* __restrict__ im = L.im[lev];
    const bool words = (w & 3) == 0 && (((size_t)im) & 3) == 0;
    for (int i = threadIdx.x; i < (TH + 6) * (FAST_LW / 4); i += 256) {
        const int r = i / (FAST_LW / 4), cw = i - r * (FAST_LW / 4);
        const int x = x0 - 4 + 4 * cw, y = y0 - 3 + r;
        uint32_t v = 0;
        if (y >= 0 && y < h) {
            if (words && x >= 0 && x + 3 < w)
                v = *reinterpret_cast<const uint32_t*>(im + (size_t)y * w + x);
            else
                for (int k = 0; k < 4; k++)
                    if (x + k >= 0 && x + k < w) v |= (uint32_t)im[(size_t)y * w + x + k] << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(&tile[r * FAST_LW + 4 * cw]) = v;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = x0 + lx, thr = L.thr[lev];
#pragma unroll
    for (int ry = 0; ry < TH / 4; ry++) {
        const int ly = wv + 4 * ry, y = y0 + ly;
        bool corner = false;
        if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
            const uint8_t* c = &tile[(ly + 3) * FAST_LW + lx + 4];
            const int v = *c, hi = v + thr, lo = v - thr;
            const int ring[16] = {c[3 * FAST_LW],      c[3 * FAST_LW + 1],  c[2 * FAST_LW + 2],  c[FAST_LW + 3],
                                  c[3],                c[-FAST_LW + 3],     c[-2 * FAST_LW + 2], c[-3 * FAST_LW + 1],
                                  c[-3 * FAST_LW],     c[-3 * FAST_LW - 1], c[-2 * FAST_LW - 2], c[-FAST_LW - 3],
                                  c[-3],               c[FAST_LW - 3],      c[2 * FAST_LW - 2],  c[3 * FAST_LW - 1]};
            unsigned br = 0, dk = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                br |= (unsigned)(ring[i] > hi) << i;
                dk |= (unsigned)(ring[i] < lo) << i;
            }
            corner = has_run10(br) || has_run10(dk);
        }
        const unsigned long long m = __ballot(corner);
        if (lx == 0 && y < h) L.mask[lev][(size_t)y * ntx + tx] = m;
    }
}
#define FAST_BATCH_TH 16
__global__ void __launch_bounds__(256) fast_detect_kernel(KfLevels L) { fast_detect_body(L, blockIdx.x); }
// the keyframes of a batch of frames (ptam_track_map_frames_batch): blockIdx.y picks the keyframe, `stride` bytes apart
__global__ void __launch_bounds__(256) fast_detect_batch_kernel(const char* __restrict__ items, size_t stride, size_t off_levels, int n_blocks) {
    const KfLevels& L = *(const KfLevels*)(items + (size_t)blockIdx.y * stride + off_levels);
    (void)n_blocks;
    fast_detect_tall_body<FAST_BATCH_TH>(L, blockIdx.x);
}
void kf_launch_detect_batch(int nb, const KfLevels& L, const void* d_items, size_t stride, size_t off_levels, hipStream_t stream) {
    int n_blocks = 0;
    for (int l = 0; l < PTAM_LEVELS; l++) n_blocks += L.ntx[l] * ((L.h[l] + FAST_BATCH_TH - 1) / FAST_BATCH_TH);
    hipLaunchKernelGGL(fast_detect_batch_kernel, dim3(n_blocks, nb), dim3(256), 0, stream, (const char*)d_items, stride, off_levels, n_blocks);
}

// ------------------------------------------------------------------------------------------------
// K2b: ceil(entries / 1024) workgroups per level, independent of each other (body: keyframe_device.h).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) fast_compact_kernel(KfLevels L, int rest) { fast_compact_body(L, blockIdx.x, rest); }

// ------------------------------------------------------------------------------------------------
// MakeKeyFrame_Rest (src/KeyFrame.cc:61-82) without the SmallBlurryImage.
//   fast_score_kernel   libCVD fast_corner_score_10: largest threshold at which the corner survives
//                       = max over the 16 ten-pixel arcs of min(ring - p) resp. min(p - ring), minus 1
//                       (closed form of the library's binary search); written into a score map
//   fast_nonmax_kernel  non-strict 3x3 suppression on the score map (non-corners score 0): a corner
//                       is kept unless an 8-neighbour scores strictly higher; kept bits -> masks
//   fast_compact_kernel (shared with K2b) orders them; shi_tomasi_kernel scores the survivors
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int arc10_max_of_min(const int d[16]) {
    int m2[16], m4[16], m8[16], best = -100000;
#pragma unroll
    for (int k = 0; k < 16; k++) m2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m8[k] = min(m4[k], m4[(k + 4) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) best = max(best, min(m8[k], m2[(k + 8) & 15]));
    return best;
}

__global__ void __launch_bounds__(256) fast_score_kernel(KfLevels L, int lev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.ncorners[lev]) return;
    const int w = L.w[lev];
    const ptam_int2 c = L.corners[lev][i];
    const uint8_t* p = L.im[lev] + (size_t)c.y * w + c.x;
    const int v = *p;
    const int ring[16] = {p[3 * w],      p[3 * w + 1],  p[2 * w + 2],  p[w + 3],  p[3],  p[-w + 3], p[-2 * w + 2], p[-3 * w + 1],
                          p[-3 * w],     p[-3 * w - 1], p[-2 * w - 2], p[-w - 3], p[-3], p[w - 3],  p[2 * w - 2],  p[3 * w - 1]};
    int up[16], dn[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        up[k] = ring[k] - v;
        dn[k] = v - ring[k];
    }
    const int s = max(arc10_max_of_min(up), arc10_max_of_min(dn)) - 1;
    L.score[lev][(size_t)c.y * w + c.x] = (uint8_t)s;   // 10 <= s <= 254 for a detected corner
}

__global__ void __launch_bounds__(256) fast_nonmax_kernel(KfLevels L, int lev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.ncorners[lev]) return;
    const int w = L.w[lev];
    const ptam_int2 c = L.corners[lev][i];
    const uint8_t* s = L.score[lev] + (size_t)c.y * w + c.x;   // corners lie >= 3 px inside: neighbours exist
    const int me = *s;
    const bool keep = s[-w - 1] <= me && s[-w] <= me && s[-w + 1] <= me && s[-1] <= me && s[1] <= me && s[w - 1] <= me &&
                      s[w] <= me && s[w + 1] <= me;
    if (keep) atomicOr(&L.mmask[lev][(size_t)c.y * L.ntx[lev] + (c.x >> 6)], 1ull << (c.x & 63));
}

// ImageProcess::ShiTomasiScoreAtPoint(im, 3, pos) src/ImageProcess.cc:20-47 for every maximal corner
__global__ void __launch_bounds__(256) shi_tomasi_kernel(KfLevels L, int lev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.nmax[lev]) return;
    const int w = L.w[lev], h = L.h[lev];
    const ptam_int2 c = L.mcorners[lev][i];
    double out = -1.0;
    if (c.x >= 10 && c.y >= 10 && c.x < w - 10 && c.y < h - 10) {   // in_image_with_border(*i, 10)  :68
        const uint8_t* im = L.im[lev];
        int xx = 0, yy = 0, xy = 0;   // sums of products of byte differences: exact in int (49 * 255^2 < 2^31)
        for (int y = c.y - 3; y <= c.y + 3; y++)
            for (int x = c.x - 3; x <= c.x + 3; x++) {
                const int dx = (int)im[(size_t)y * w + x + 1] - (int)im[(size_t)y * w + x - 1];
                const int dy = (int)im[(size_t)(y + 1) * w + x] - (int)im[(size_t)(y - 1) * w + x];
                xx += dx * dx;
                yy += dy * dy;
                xy += dx * dy;
            }
        const double dXX = (double)xx / (2.0 * 49), dYY = (double)yy / (2.0 * 49), dXY = (double)xy / (2.0 * 49);
        out = 0.5 * (dXX + dYY - sqrt((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY)));
    }
    L.st[lev][i] = out;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int kf_alloc(ptam_ctx* ctx, int w, int h, ptam_kf** out) {
    ptam_kf* kf = new ptam_kf();
    std::memset(kf, 0, sizeof *kf);
    kf->device = ctx->device;
    size_t px = 0, ent = 0, rows = 0;
    int blocks = 0;
    static const int thr[PTAM_LEVELS] = {10, 15, 15, 10};   // src/KeyFrame.cc:35-42
    for (int l = 0; l < PTAM_LEVELS; l++) {
        kf->L.w[l] = l ? kf->L.w[l - 1] / 2 : w;
        kf->L.h[l] = l ? kf->L.h[l - 1] / 2 : h;
        kf->L.thr[l] = thr[l];
        kf->L.ntx[l] = (kf->L.w[l] + FAST_TW - 1) / FAST_TW;
        kf->L.block_begin[l] = blocks;
        blocks += kf->L.ntx[l] * ((kf->L.h[l] + FAST_TH - 1) / FAST_TH);
        px += ((size_t)kf->L.w[l] * kf->L.h[l] + 255) & ~(size_t)255;
        ent += (size_t)kf->L.ntx[l] * kf->L.h[l];
        rows += kf->L.h[l];
    }
    kf->n_blocks = blocks;
    kf->bytes_px = px;
    // one allocation: pixels | masks | corners (worst case = every pixel) | rowluts | counts |
    //                 score maps | max masks | max corners | Shi-Tomasi scores   (MakeKeyFrame_Rest)
    const size_t b_mask = ent * 8, b_corn = px * sizeof(ptam_int2), b_lut = rows * 4 + 64;
    const size_t off_rest = (px + b_mask + b_corn + b_lut + 64 + 255) & ~(size_t)255;
    kf->bytes_total = off_rest + px + b_mask + b_corn + px * 8 + 64;
    hipError_t e = hipMalloc(&kf->base, kf->bytes_total);
    if (e != hipSuccess) {
        ptam_set_error("hipMalloc(%zu) failed: %s", kf->bytes_total, hipGetErrorString(e));
        delete kf;
        return PTAM_E_HIP;
    }
    char* p = (char*)kf->base;
    for (int l = 0; l < PTAM_LEVELS; l++) {
        kf->L.im[l] = (uint8_t*)p;
        p += ((size_t)kf->L.w[l] * kf->L.h[l] + 255) & ~(size_t)255;
    }
    for (int l = 0; l < PTAM_LEVELS; l++) {
        kf->L.mask[l] = (unsigned long long*)p;
        p += (size_t)kf->L.ntx[l] * kf->L.h[l] * 8;
    }
    for (int l = 0; l < PTAM_LEVELS; l++) {
        kf->L.corners[l] = (ptam_int2*)p;
        p += (((size_t)kf->L.w[l] * kf->L.h[l] + 255) & ~(size_t)255) * sizeof(ptam_int2);
    }
    for (int l = 0; l < PTAM_LEVELS; l++) {
        kf->L.rowlut[l] = (int*)p;
        p += (size_t)kf->L.h[l] * 4;
    }
    p = (char*)(((uintptr_t)p + 63) & ~(uintptr_t)63);
    kf->L.ncorners = (int*)p;
    kf->L.nmax = kf->L.ncorners + PTAM_LEVELS;
    p = (char*)kf->base + off_rest;
    kf->off_rest_clear = off_rest;
    for (int l = 0; l < PTAM_LEVELS; l++) {
        kf->L.score[l] = (uint8_t*)p;
        p += ((size_t)kf->L.w[l] * kf->L.h[l] + 255) & ~(size_t)255;
    }
    for (int l = 0; l < PTAM_LEVELS; l++) {
        kf->L.mmask[l] = (unsigned long long*)p;
        p += (size_t)kf->L.ntx[l] * kf->L.h[l] * 8;
    }
    kf->bytes_rest_clear = (size_t)(p - ((char*)kf->base + off_rest));
    for (int l = 0; l < PTAM_LEVELS; l++) {
        kf->L.mcorners[l] = (ptam_int2*)p;
        p += (((size_t)kf->L.w[l] * kf->L.h[l] + 255) & ~(size_t)255) * sizeof(ptam_int2);
    }
    for (int l = 0; l < PTAM_LEVELS; l++) {
        kf->L.st[l] = (double*)p;
        p += (((size_t)kf->L.w[l] * kf->L.h[l] + 255) & ~(size_t)255) * sizeof(double);
    }
    *out = kf;
    return PTAM_OK;
}

// the pyramid's arguments and grid for a new image in this keyframe; marks everything derived from the old one stale
void kf_lite_begin(ptam_kf* kf, const uint8_t* d_src, PyrArgs* a_out, int* gx, int* gy) {
    for (int l = 0; l < PTAM_LEVELS; l++) kf->implane_valid[l] = 0;   // new corners: the in-plane cache is stale
    PyrArgs& a = *a_out;
    a.src = d_src;
    for (int l = 0; l < PTAM_LEVELS; l++) {
        a.lv[l] = kf->L.im[l];
        a.w[l] = kf->L.w[l];
        a.h[l] = kf->L.h[l];
    }
    const int nbx = (a.w[0] + 7) / 8, nby = (a.h[0] + 7) / 8;
    *gx = (nbx + 63) / 64;
    *gy = (nby + 3) / 4;
    kf->counts_valid = 0;
    kf->rest_valid = 0;
}
void kf_launch_detect(ptam_kf* kf, hipStream_t stream) { hipLaunchKernelGGL(fast_detect_kernel, dim3(kf->n_blocks), dim3(256), 0, stream, kf->L); }

static int kf_run(ptam_ctx* ctx, ptam_kf* kf, const uint8_t* d_src, hipStream_t stream) {
    PyrArgs a;
    int gx, gy;
    kf_lite_begin(kf, d_src, &a, &gx, &gy);
    dim3 blk(64, 4), grd(gx, gy);
    if (ctx->halfsample == PTAM_HALFSAMPLE_T)
        hipLaunchKernelGGL(pyramid_kernel<PTAM_HALFSAMPLE_T>, grd, blk, 0, stream, a);
    else
        hipLaunchKernelGGL(pyramid_kernel<PTAM_HALFSAMPLE_R>, grd, blk, 0, stream, a);
    hipLaunchKernelGGL(fast_detect_kernel, dim3(kf->n_blocks), dim3(256), 0, stream, kf->L);
    hipLaunchKernelGGL(fast_compact_kernel, dim3(fast_compact_blocks(kf->L)), dim3(1024), 0, stream, kf->L, 0);
    HIP_TRY(hipGetLastError());
    kf->counts_valid = 0;
    kf->rest_valid = 0;
    return PTAM_OK;
}

// MakeKeyFrame_Lite of a device-resident frame on a stream of the caller's choosing (track_internal.h)
int kf_make_lite_on(ptam_ctx* ctx, ptam_kf* kf, const uint8_t* d_im, hipStream_t stream) { return kf_run(ctx, kf, d_im, stream); }

int kf_fetch_counts(ptam_ctx* ctx, const ptam_kf* kf_c) {
    ptam_kf* kf = const_cast<ptam_kf*>(kf_c);
    if (kf->counts_valid) return PTAM_OK;
    HIP_TRY(hipMemcpyAsync(kf->n_corners, kf->L.ncorners, sizeof kf->n_corners, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    kf->counts_valid = 1;
    return PTAM_OK;
}

extern "C" {

int ptam_kf_create(ptam_ctx* ctx, int width, int height, ptam_kf** out) {
    ARG_TRY(ctx && out && width >= 8 && height >= 8);
    HIP_TRY(hipSetDevice(ctx->device));
    return kf_alloc(ctx, width, height, out);
}

int ptam_kf_destroy(ptam_kf* kf) {
    if (!kf) return PTAM_OK;
    hipSetDevice(kf->device);
    hipDeviceSynchronize();
    hipFree(kf->base);
    for (int l = 0; l < PTAM_LEVELS; l++)
        if (kf->implane[l]) hipFree(kf->implane[l]);
    delete kf;
    return PTAM_OK;
}

int ptam_make_keyframe_lite(ptam_ctx* ctx, ptam_kf* kf, const uint8_t* im, int stride) {
    ARG_TRY(ctx && kf && im && stride >= kf->L.w[0]);
    HIP_TRY(hipSetDevice(ctx->device));
    // copy(im, aLevels[0].im)  src/KeyFrame.cc:20-21 — straight into the keyframe's level 0
    HIP_TRY(hipMemcpy2DAsync(kf->L.im[0], kf->L.w[0], im, stride, kf->L.w[0], kf->L.h[0], hipMemcpyHostToDevice,
                             ctx->stream));
    return kf_run(ctx, kf, kf->L.im[0], ctx->stream);
}

int ptam_make_keyframe_lite_dev(ptam_ctx* ctx, ptam_kf* kf, const uint8_t* d_im) {
    ARG_TRY(ctx && kf && d_im);
    HIP_TRY(hipSetDevice(ctx->device));
    return kf_run(ctx, kf, d_im, ctx->stream);   // the pyramid kernel also copies src -> level 0
}

int ptam_make_keyframe_rest(ptam_ctx* ctx, ptam_kf* kf) {
    ARG_TRY(ctx && kf);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = kf_fetch_counts(ctx, kf);   // launch sizes (the mapmaker thread calls this, not the frame loop)
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync((char*)kf->base + kf->off_rest_clear, 0, kf->bytes_rest_clear, ctx->stream));
    for (int l = 0; l < PTAM_LEVELS; l++) {
        const int n = kf->n_corners[l];
        if (n > 0) hipLaunchKernelGGL(fast_score_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, kf->L, l);
    }
    for (int l = 0; l < PTAM_LEVELS; l++) {
        const int n = kf->n_corners[l];
        if (n > 0) hipLaunchKernelGGL(fast_nonmax_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, kf->L, l);
    }
    hipLaunchKernelGGL(fast_compact_kernel, dim3(fast_compact_blocks(kf->L)), dim3(1024), 0, ctx->stream, kf->L, 1);
    for (int l = 0; l < PTAM_LEVELS; l++) {
        const int n = kf->n_corners[l];   // upper bound of the maximal corners; the kernel checks nmax
        if (n > 0) hipLaunchKernelGGL(shi_tomasi_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, kf->L, l);
    }
    HIP_TRY(hipGetLastError());
    kf->rest_valid = 0;
    return PTAM_OK;
}

static int kf_fetch_rest(ptam_ctx* ctx, const ptam_kf* kf_c) {
    ptam_kf* kf = const_cast<ptam_kf*>(kf_c);
    if (kf->rest_valid) return PTAM_OK;
    HIP_TRY(hipMemcpyAsync(kf->n_max, kf->L.nmax, sizeof kf->n_max, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    kf->rest_valid = 1;
    return PTAM_OK;
}

int ptam_kf_rest_info(ptam_ctx* ctx, const ptam_kf* kf, int level, int* n_max_corners) {
    ARG_TRY(ctx && kf && level >= 0 && level < PTAM_LEVELS && n_max_corners);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = kf_fetch_rest(ctx, kf);
    if (rc) return rc;
    *n_max_corners = kf->n_max[level];
    return PTAM_OK;
}

int ptam_kf_read_rest(ptam_ctx* ctx, const ptam_kf* kf, int level, ptam_int2* max_corners, double* st_scores) {
    ARG_TRY(ctx && kf && level >= 0 && level < PTAM_LEVELS);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = kf_fetch_rest(ctx, kf);
    if (rc) return rc;
    const int n = kf->n_max[level];
    if (n > 0 && max_corners)
        HIP_TRY(hipMemcpyAsync(max_corners, kf->L.mcorners[level], (size_t)n * sizeof(ptam_int2), hipMemcpyDeviceToHost,
                               ctx->stream));
    if (n > 0 && st_scores)
        HIP_TRY(hipMemcpyAsync(st_scores, kf->L.st[level], (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

int ptam_kf_clone(ptam_ctx* ctx, const ptam_kf* src, ptam_kf** out) {
    ARG_TRY(ctx && src && out);
    HIP_TRY(hipSetDevice(ctx->device));
    ptam_kf* kf = nullptr;
    int rc = kf_alloc(ctx, src->L.w[0], src->L.h[0], &kf);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(kf->base, src->base, src->bytes_total, hipMemcpyDeviceToDevice, ctx->stream));
    kf->counts_valid = 0;
    kf->rest_valid = 0;
    *out = kf;
    return PTAM_OK;
}

int ptam_kf_level_info(ptam_ctx* ctx, const ptam_kf* kf, int level, int* w, int* h, int* n_corners) {
    ARG_TRY(ctx && kf && level >= 0 && level < PTAM_LEVELS);
    HIP_TRY(hipSetDevice(ctx->device));
    if (w) *w = kf->L.w[level];
    if (h) *h = kf->L.h[level];
    if (n_corners) {
        int rc = kf_fetch_counts(ctx, kf);
        if (rc) return rc;
        *n_corners = kf->n_corners[level];
    }
    return PTAM_OK;
}

int ptam_kf_read_level(ptam_ctx* ctx, const ptam_kf* kf, int level, uint8_t* px, ptam_int2* corners,
                       int32_t* rowlut) {
    ARG_TRY(ctx && kf && level >= 0 && level < PTAM_LEVELS);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = kf_fetch_counts(ctx, kf);
    if (rc) return rc;
    const int w = kf->L.w[level], h = kf->L.h[level], n = kf->n_corners[level];
    if (px) HIP_TRY(hipMemcpyAsync(px, kf->L.im[level], (size_t)w * h, hipMemcpyDeviceToHost, ctx->stream));
    if (corners && n > 0)
        HIP_TRY(hipMemcpyAsync(corners, kf->L.corners[level], (size_t)n * sizeof(ptam_int2), hipMemcpyDeviceToHost,
                               ctx->stream));
    if (rowlut) HIP_TRY(hipMemcpyAsync(rowlut, kf->L.rowlut[level], (size_t)h * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    return PTAM_OK;
}

}   // extern "C"

// every kernel of this file resolved once, when a context is created: the first launch of a kernel otherwise pays for
// loading the code object / resolving the function — 10-28 ms in the middle of the first frame or the first adjustment
void kf_preload_kernels() {
    ptam_preload((const void*)fast_detect_kernel);
    ptam_preload((const void*)fast_compact_kernel);
    ptam_preload((const void*)fast_detect_batch_kernel);
    ptam_preload((const void*)fast_score_kernel);
    ptam_preload((const void*)fast_nonmax_kernel);
    ptam_preload((const void*)shi_tomasi_kernel);
    ptam_preload((const void*)pyramid_kernel<PTAM_HALFSAMPLE_R>);
    ptam_preload((const void*)pyramid_kernel<PTAM_HALFSAMPLE_T>);
}
