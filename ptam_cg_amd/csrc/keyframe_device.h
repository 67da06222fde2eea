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
