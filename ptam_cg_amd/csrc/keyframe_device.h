// keyframe_device.h — device bodies of the keyframe kernels that other translation units fuse into their own launches
// (trackmap.hip runs the pyramid beside the PVS pass and the corner compaction beside the set choice): keyframe.hip wraps
// the same bodies as stand-alone kernels.
#pragma once
#include "common.h"
#include "keyframe.h"

#define FAST_TW 64
#define FAST_TH 4
#define FAST_LW (FAST_TW + 6 + 2)   // padded row pitch (72 bytes)

// ------------------------------------------------------------------------------------------------
// K1: halfSample cascade.  Thread = one 8x8 block of L0 -> 4x4 of L1, 2x2 of L2, 1 of L3.
// Variant R (default) = libCVD SSE2 byte path: vertical pavgb then horizontal pavgw;
// variant T = (a+b+c+d)/4 truncating.  
// ------------------------------------------------------------------------------------------------
template <int VARIANT>
__device__ __forceinline__ int half4(int a, int b, int c, int d) {
    if (VARIANT == PTAM_HALFSAMPLE_T) return (a + b + c + d) >> 2;   // operands >= 0: shift == C division
    const int v1 = (a + c + 1) >> 1, v2 = (b + d + 1) >> 1;
    return (v1 + v2 + 1) >> 1;
}

struct PyrArgs {
    const uint8_t* src;   // source image (device), stride == w0
    uint8_t* lv[4];       // level images; lv[0] is written iff src != lv[0]
    int w[4], h[4];
};

template <int VARIANT>
__device__ __forceinline__ void pyramid_body(const PyrArgs& a, int bx, int by) {   // (bx, by): 8x8 block column / row
    const int x0 = bx * 8, y0 = by * 8;
    if (x0 >= a.w[0] || y0 >= a.h[0]) return;
    const bool full = (x0 + 8 <= a.w[0]) && (y0 + 8 <= a.h[0]) && ((a.w[0] & 7) == 0);
    uint8_t p[8][8];
    if (full) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint2 v = *reinterpret_cast<const uint2*>(a.src + (size_t)(y0 + r) * a.w[0] + x0);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                p[r][c] = (v.x >> (8 * c)) & 0xff;
                p[r][4 + c] = (v.y >> (8 * c)) & 0xff;
            }
        }
        if (a.src != a.lv[0]) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint2 v;
                v.x = p[r][0] | (p[r][1] << 8) | (p[r][2] << 16) | ((unsigned)p[r][3] << 24);
                v.y = p[r][4] | (p[r][5] << 8) | (p[r][6] << 16) | ((unsigned)p[r][7] << 24);
                *reinterpret_cast<uint2*>(a.lv[0] + (size_t)(y0 + r) * a.w[0] + x0) = v;
            }
        }
    } else {
        for (int r = 0; r < 8; r++)
            for (int c = 0; c < 8; c++) {
                const int x = x0 + c, y = y0 + r;
                const bool in = x < a.w[0] && y < a.h[0];
                p[r][c] = in ? a.src[(size_t)y * a.w[0] + x] : 0;
                if (in && a.src != a.lv[0]) a.lv[0][(size_t)y * a.w[0] + x] = p[r][c];
            }
    }
    // L1: 4x4
    uint8_t q[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            q[r][c] = (uint8_t)half4<VARIANT>(p[2 * r][2 * c], p[2 * r][2 * c + 1], p[2 * r + 1][2 * c], p[2 * r + 1][2 * c + 1]);
    const int x1 = bx * 4, y1 = by * 4;
    if (full) {
#pragma unroll
        for (int r = 0; r < 4; r++)
            *reinterpret_cast<uint32_t*>(a.lv[1] + (size_t)(y1 + r) * a.w[1] + x1) =
                q[r][0] | (q[r][1] << 8) | (q[r][2] << 16) | ((unsigned)q[r][3] << 24);
    } else {
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++)
                if (x1 + c < a.w[1] && y1 + r < a.h[1]) a.lv[1][(size_t)(y1 + r) * a.w[1] + x1 + c] = q[r][c];
    }
    // L2: 2x2
    uint8_t s[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 2; c++)
            s[r][c] = (uint8_t)half4<VARIANT>(q[2 * r][2 * c], q[2 * r][2 * c + 1], q[2 * r + 1][2 * c], q[2 * r + 1][2 * c + 1]);
    const int x2 = bx * 2, y2 = by * 2;
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++)
            if (x2 + c < a.w[2] && y2 + r < a.h[2]) a.lv[2][(size_t)(y2 + r) * a.w[2] + x2 + c] = s[r][c];
    // L3
    if (bx < a.w[3] && by < a.h[3])
        a.lv[3][(size_t)by * a.w[3] + bx] = (uint8_t)half4<VARIANT>(s[0][0], s[0][1], s[1][0], s[1][1]);
}


// ------------------------------------------------------------------------------------------------
// (round 4) K1 + K2a in ONE launch: a FAST tile of level l computes the level-l pixels it needs — its 64 x 4 interior and the
// 3-pixel halo — straight from the SOURCE frame (the halfSample cascade of the (2^l x 2^l) block under each pixel, in registers,
// the same half4 chain as pyramid_body: the same bytes), writes its interior to the level image and tests it.  Halo pixels are
// computed twice (by the neighbour too) and written once; levels 1-3 re-read the frame 1.3 / 1.0 / 0.8 MB out of the L2.  The
// pyramid kernel and its boundary (4.9 + 2.3 us of a tracked frame) are gone; the tiles of levels 2-3, which do the extra
// arithmetic, are 10 % of the launch's workgroups and run beside the level-0 tiles.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool kf_has_run10(unsigned m16) {
    const unsigned d = m16 | (m16 << 16);
    const unsigned a = d & (d >> 1);
    const unsigned b = a & (a >> 2);
    const unsigned c = b & (b >> 4);
    return (c & (a >> 8) & 0xffffu) != 0;
}
// pixel (x, y) of level `lev` (1..3) from the level-0 image: the cascade over its (1 << lev)^2 block
template <int VARIANT>
__device__ __forceinline__ int kf_pyr_pixel(const uint8_t* __restrict__ src, int w0, int lev, int x, int y, bool aligned) {
    if (lev == 1) {
        const uint8_t* p = src + (size_t)(2 * y) * w0 + 2 * x;
        return half4<VARIANT>(p[0], p[1], p[w0], p[w0 + 1]);
    }
    if (lev == 2) {
        uint8_t q[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint8_t* p = src + (size_t)(4 * y + r) * w0 + 4 * x;
            if (aligned) {
                const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
                for (int c = 0; c < 4; c++) q[r][c] = (v >> (8 * c)) & 0xff;
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) q[r][c] = p[c];
            }
        }
        int s[2][2];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 2; c++) s[r][c] = half4<VARIANT>(q[2 * r][2 * c], q[2 * r][2 * c + 1], q[2 * r + 1][2 * c], q[2 * r + 1][2 * c + 1]);
        return half4<VARIANT>(s[0][0], s[0][1], s[1][0], s[1][1]);
    }
    uint8_t p8[8][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint8_t* p = src + (size_t)(8 * y + r) * w0 + 8 * x;
        if (aligned) {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                p8[r][c] = (v.x >> (8 * c)) & 0xff;
                p8[r][4 + c] = (v.y >> (8 * c)) & 0xff;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; c++) p8[r][c] = p[c];
        }
    }
    uint8_t q[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) q[r][c] = (uint8_t)half4<VARIANT>(p8[2 * r][2 * c], p8[2 * r][2 * c + 1], p8[2 * r + 1][2 * c], p8[2 * r + 1][2 * c + 1]);
    int s[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 2; c++) s[r][c] = half4<VARIANT>(q[2 * r][2 * c], q[2 * r][2 * c + 1], q[2 * r + 1][2 * c], q[2 * r + 1][2 * c + 1]);
    return half4<VARIANT>(s[0][0], s[0][1], s[1][0], s[1][1]);
}
// W neighbouring pixels of level log2(N), packed low byte first, from their N rows x W*N bytes of the frame (aligned rows)
template <int VARIANT, int N, int W>
__device__ __forceinline__ uint32_t kf_cascade_word(uint8_t (&p)[N][W * N]) {
    if constexpr (N == 1) {
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < W; c++) v |= (uint32_t)p[0][c] << (8 * c);
        return v;
    } else {
        uint8_t q[N / 2][W * N / 2];
#pragma unroll
        for (int r = 0; r < N / 2; r++)
#pragma unroll
            for (int c = 0; c < W * N / 2; c++) q[r][c] = (uint8_t)half4<VARIANT>(p[2 * r][2 * c], p[2 * r][2 * c + 1], p[2 * r + 1][2 * c], p[2 * r + 1][2 * c + 1]);
        return kf_cascade_word<VARIANT, N / 2, W>(q);
    }
}
template <int VARIANT, int N, int W>
__device__ __forceinline__ uint32_t kf_pyr_word(const uint8_t* __restrict__ src, int w0, int x, int y) {
    uint8_t p[N][W * N];
    const uint8_t* base = src + (size_t)(N * y) * w0 + N * x;
#pragma unroll
    for (int r = 0; r < N; r++) {
        if constexpr (W * N == 4) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(base);
#pragma unroll
            for (int c = 0; c < 4; c++) p[r][c] = (v >> (8 * c)) & 0xff;
        } else if constexpr (W * N == 8) {
            const uint2 v = *reinterpret_cast<const uint2*>(base + (size_t)r * w0);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                p[r][c] = (v.x >> (8 * c)) & 0xff;
                p[r][4 + c] = (v.y >> (8 * c)) & 0xff;
            }
        } else {
            static_assert(W * N == 16, "one 16-byte load per row");
            const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)r * w0);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                p[r][c] = (v.x >> (8 * c)) & 0xff;
                p[r][4 + c] = (v.y >> (8 * c)) & 0xff;
                p[r][8 + c] = (v.z >> (8 * c)) & 0xff;
                p[r][12 + c] = (v.w >> (8 * c)) & 0xff;
            }
        }
    }
    return kf_cascade_word<VARIANT, N, W>(p);
}
template <int VARIANT>
__device__ __forceinline__ void kf_fused_tile_body(const PyrArgs& a, const KfLevels& L, int bx) {   // a 256-thread workgroup = one FAST tile
    __shared__ __attribute__((aligned(4))) uint8_t ktile[(FAST_TH + 6) * FAST_LW];
    static_assert(FAST_LW == FAST_TW + 8, "tile pitch = 18 words");
    int lev = 0;
#pragma unroll
    for (int l = 1; l < PTAM_LEVELS; l++)
        if (bx >= L.block_begin[l]) lev = l;
    const int b = bx - L.block_begin[lev];
    const int w = L.w[lev], h = L.h[lev], ntx = L.ntx[lev];
    const int tx = b % ntx, ty = b / ntx;
    const int x0 = tx * FAST_TW, y0 = ty * FAST_TH;
    const int w0 = a.w[0];
    uint8_t* __restrict__ out = a.lv[lev];
    const bool write = lev > 0 || a.src != a.lv[0];
    // the tile as 10 rows x 18 words from column x0 - 4: ONE word per thread, computed from 1 / 2 / 4 aligned loads of the
    // frame (widths that are multiples of 32 — every level's width a multiple of 4 — and 16-byte-aligned images; any other
    // frame goes pixel by pixel below)
    const bool words = (w0 & 31) == 0 && (((size_t)a.src) & 15) == 0 && (((size_t)out) & 3) == 0;   // (level widths: multiples of 4)
    if (words && lev < 3) {
        const int i = threadIdx.x;
        if (i < (FAST_TH + 6) * (FAST_LW / 4)) {
            const int r = i / (FAST_LW / 4), cw = i - r * (FAST_LW / 4);
            const int x = x0 - 4 + 4 * cw, y = y0 - 3 + r;
            uint32_t v = 0;
            if (y >= 0 && y < h && x >= 0 && x < w) {
                v = lev == 0   ? kf_pyr_word<VARIANT, 1, 4>(a.src, w0, x, y)
                    : lev == 1 ? kf_pyr_word<VARIANT, 2, 4>(a.src, w0, x, y)
                               : kf_pyr_word<VARIANT, 4, 4>(a.src, w0, x, y);
                if (write && cw >= 1 && cw <= FAST_TW / 4 && r >= 3 && r < 3 + FAST_TH) *reinterpret_cast<uint32_t*>(out + (size_t)y * w + x) = v;
            }
            *reinterpret_cast<uint32_t*>(&ktile[r * FAST_LW + 4 * cw]) = v;
        }
    } else if (words) {
        // level 3: pairs of pixels (8 rows x 16 bytes of the frame each), two passes — the 30 tiles of the level start first
        for (int i = threadIdx.x; i < (FAST_TH + 6) * (FAST_LW / 2); i += 256) {
            const int r = i / (FAST_LW / 2), ch = i - r * (FAST_LW / 2);
            const int x = x0 - 4 + 2 * ch, y = y0 - 3 + r;
            uint32_t v = 0;
            if (y >= 0 && y < h && x >= 0 && x < w) {
                v = kf_pyr_word<VARIANT, 8, 2>(a.src, w0, x, y);
                if (ch >= 2 && ch < 2 + FAST_TW / 2 && r >= 3 && r < 3 + FAST_TH) *reinterpret_cast<uint16_t*>(out + (size_t)y * w + x) = (uint16_t)v;
            }
            *reinterpret_cast<uint16_t*>(&ktile[r * FAST_LW + 2 * ch]) = (uint16_t)v;
        }
    } else {
        const bool aligned = (w0 & 7) == 0 && (((size_t)a.src) & 7) == 0;
        for (int i = threadIdx.x; i < (FAST_TH + 6) * (FAST_TW + 6); i += 256) {
            const int r = i / (FAST_TW + 6), c = i - r * (FAST_TW + 6);
            const int x = x0 - 3 + c, y = y0 - 3 + r;
            int v = 0;
            if (x >= 0 && x < w && y >= 0 && y < h) {
                v = lev == 0 ? (int)a.src[(size_t)y * w0 + x] : kf_pyr_pixel<VARIANT>(a.src, w0, lev, x, y, aligned);
                if (write && c >= 3 && c < 3 + FAST_TW && r >= 3 && r < 3 + FAST_TH) out[(size_t)y * w + x] = (uint8_t)v;   // the tile's own pixels
            }
            ktile[r * FAST_LW + c + 1] = (uint8_t)v;
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = x0 + lx, y = y0 + ly;
    bool corner = false;
    if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
        const uint8_t* c = &ktile[(ly + 3) * FAST_LW + lx + 4];
        const int v = *c, hi = v + L.thr[lev], lo = v - L.thr[lev];
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
        corner = kf_has_run10(br) || kf_has_run10(dk);
    }
    const unsigned long long m = __ballot(corner);
    if (lx == 0 && y < h) L.mask[lev][(size_t)y * ntx + tx] = m;
}

// ------------------------------------------------------------------------------------------------
// K2b: raster-ordered corner lists + row LUTs from the (row, tile) bit masks.  Entries e = y*ntx + tx in raster order, ONE
// entry per thread, ceil(E / 1024) workgroups per level (640x480: 5 + 2 + 1 + 1).  A workgroup needs the number of corners
// in front of its slice: it counts them itself — the masks of the whole level are 38 KB at most, a preceding slice is one
// more 8-byte load per thread, issued together with the thread's own — so the workgroups of a level never talk to each
// other (no look-back chain, no second launch), and the block-wide exclusive scan of the slice gives the offsets.
// rowlut[y] = offset of entry (y, 0) = index of the first corner with row >= y (src/KeyFrame.cc:46-52).
// (The first version ran ONE workgroup per level with five consecutive entries per thread: 6.6 us for 640x480.)
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int fast_compact_blocks(const KfLevels& L) {
    int n = 0;
    for (int l = 0; l < PTAM_LEVELS; l++) n += (L.h[l] * L.ntx[l] + 1023) / 1024;
    return n;
}
__device__ __forceinline__ void fast_compact_body(const KfLevels& L, int blk, int rest) {   // a 1024-thread workgroup
    __shared__ int wsum[16], wbase[16];
    int lev = 0, b = blk;
    for (; lev < PTAM_LEVELS - 1; lev++) {
        const int nb = (L.h[lev] * L.ntx[lev] + 1023) / 1024;
        if (b < nb) break;
        b -= nb;
    }
    const int ntx = L.ntx[lev], E = L.h[lev] * ntx;
    const unsigned long long* __restrict__ mask = rest ? L.mmask[lev] : L.mask[lev];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int e = b * 1024 + tid;
    unsigned long long m = e < E ? mask[e] : 0ull;
    int before = 0;
    for (int k = 0; k < b; k++) before += __popcll(mask[k * 1024 + tid]);
    const int cnt = __popcll(m);
    const int incl = wave_incl_scan_i32(cnt);
    before = wave_sum_i32(before);
    if (lane == 63) wsum[wid] = incl;
    if (lane == 0) wbase[wid] = before;
    __syncthreads();
    int base = 0, off = 0, slice = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        base += wbase[i];
        off += i < wid ? wsum[i] : 0;
        slice += wsum[i];
    }
    off += base + incl - cnt;
    ptam_int2* __restrict__ out = rest ? L.mcorners[lev] : L.corners[lev];
    if (e < E) {
        const int y = e / ntx, tx = e - y * ntx;
        if (tx == 0 && !rest) L.rowlut[lev][y] = off;
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            out[off++] = ptam_int2{tx * FAST_TW + bit, y};
        }
    }
    if (tid == 0 && (b + 1) * 1024 >= E) (rest ? L.nmax : L.ncorners)[lev] = base + slice;
}
