// patch_device.h — the per-wave routines of the patch pipeline (one wave = one 8x8 patch, lane = pixel), shared by the batch
// kernels of patch.hip and the fused search kernel of the resident TrackMap chain (trackmap.hip).
#pragma once
#include "common.h"
#include "keyframe.h"
#include "track_internal.h"

// fp64 primitives that are never contracted into FMAs (hipcc's __dmul_rn / __dadd_rn / __dsub_rn are plain operators
// under -ffp-contract=fast and WOULD be fused): each rounds on its own, like the reference's x86-64 build
__device__ __forceinline__ double nc_mul(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double nc_add(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double nc_sub(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float nc_mulf(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float nc_addf(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

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

// Four candidates per pass: 16 lanes per candidate, 4 consecutive window pixels per lane (one 32-bit load), the three sums
// by v_dot4_u32_u8 against the lane's packed template pixels and four DPP row steps — ~45 instructions per four candidates
// instead of ~80 per candidate with a lane per pixel and three whole-wave sums (integer arithmetic: the same numbers, bit
// for bit).  wave_pack_template4: my four template pixels, packed like the image word (row sub / 2, columns 4 (sub % 2) .. + 3,
// sub = lane % 16).  wave_zmssd4: lane 16 k + 15 returns candidate k's score (PTAM_MAX_SSD + 1 outside the border, like wave_zmssd).
__device__ __forceinline__ unsigned wave_pack_template4(int T, int lane) {
    const int sub = lane & 15, tb = (sub >> 1) * 8 + (sub & 1) * 4;
    return (unsigned)__shfl(T, tb, 64) | ((unsigned)__shfl(T, tb + 1, 64) << 8) | ((unsigned)__shfl(T, tb + 2, 64) << 16) |
           ((unsigned)__shfl(T, tb + 3, 64) << 24);
}
// The search region of a FindPatchCoarse call staged in LDS by its wave (SW_SIDE x SW_SIDE bytes, origin (x0, y0) inside the
// image): every candidate window then comes out of LDS instead of costing a dependent global round trip per four candidates.
#define SW_SIDE 32
#define SW_LIST 128                                 // candidate list of the lane-per-candidate path (int2 each)
#define SW_BYTES (SW_SIDE * SW_SIDE + 16 + SW_LIST * 8)
struct SearchWin {
    const unsigned* lds;   // nullptr: score from global memory
    int x0, y0;
};
__device__ __forceinline__ int wave_zmssd4(const uint8_t* __restrict__ im, int w, int h, const int cx[4], const int cy[4], int n,
                                           unsigned T4, int tsum, int tsumsq, int lane, const SearchWin sw = SearchWin{nullptr, 0, 0}) {
    typedef unsigned u32_unaligned __attribute__((aligned(1)));
    const int sub = lane & 15, grp = lane >> 4;
    const int mx = grp == 0 ? cx[0] : grp == 1 ? cx[1] : grp == 2 ? cx[2] : cx[3];
    const int my = grp == 0 ? cy[0] : grp == 1 ? cy[1] : grp == 2 ? cy[2] : cy[3];
    const bool inb = grp < n && mx >= 4 && my >= 4 && mx < w - 4 && my < h - 4;
    unsigned I4 = 0;
    if (sw.lds) {
        // four bytes at an arbitrary byte offset of the window: two aligned words and a byte shift
        const int off = inb ? (my - 4 + (sub >> 1) - sw.y0) * SW_SIDE + (mx - 4 + 4 * (sub & 1) - sw.x0) : 0;
        const unsigned lo = sw.lds[off >> 2], hi = sw.lds[(off >> 2) + 1];
        I4 = inb ? __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(off & 3)) : 0u;
    } else if (inb)
        I4 = *(const u32_unaligned*)(im + (size_t)(my - 4 + (sub >> 1)) * w + (mx - 4 + 4 * (sub & 1)));
    int s1 = (int)__builtin_amdgcn_udot4(I4, 0x01010101u, 0u, false);
    int s2 = (int)__builtin_amdgcn_udot4(I4, I4, 0u, false);
    int s3 = (int)__builtin_amdgcn_udot4(I4, T4, 0u, false);
    s1 += dpp_row_shr0_i32<1>(s1), s2 += dpp_row_shr0_i32<1>(s2), s3 += dpp_row_shr0_i32<1>(s3);
    s1 += dpp_row_shr0_i32<2>(s1), s2 += dpp_row_shr0_i32<2>(s2), s3 += dpp_row_shr0_i32<2>(s3);
    s1 += dpp_row_shr0_i32<4>(s1), s2 += dpp_row_shr0_i32<4>(s2), s3 += dpp_row_shr0_i32<4>(s3);
    s1 += dpp_row_shr0_i32<8>(s1), s2 += dpp_row_shr0_i32<8>(s2), s3 += dpp_row_shr0_i32<8>(s3);
    return inb ? zmssd_finish(tsum, s1, s2, tsumsq, s3) : PTAM_MAX_SSD + 1;   // (lane 15 of each 16-lane row holds its candidate's)
}

// PatchFinder::FindPatchCoarse (src/PatchFinder.cc:160-211) by one wave: lane = pixel of the 8x8 window, T = this lane's
// template pixel.  enabled = false: the query is not searched (bad template).
// wave minimum of an unsigned key, in every lane (row shifts keep a lane's own value where there is no source)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)dpp_row_shr_i32<1>((int)v, (int)v));
    v = min(v, (unsigned)dpp_row_shr_i32<2>((int)v, (int)v));
    v = min(v, (unsigned)dpp_row_shr_i32<4>((int)v, (int)v));
    v = min(v, (unsigned)dpp_row_shr_i32<8>((int)v, (int)v));
    v = min(v, (unsigned)dpp_bcast_i32<0x142, 0xA>((int)v, (int)v));
    v = min(v, (unsigned)dpp_bcast_i32<0x143, 0xC>((int)v, (int)v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// ONE candidate per lane, its whole 8 x 8 window out of the staged region: per row three aligned words, two byte shifts, six
// v_dot4 against the row's template words (Tw[16]: wave-uniform, row r = words 2 r, 2 r + 1) — ~110 instructions for up to 64
// candidates, against ~80 per FOUR candidates in the 16-lanes-per-candidate form, whose passes were a serial chain of scalar
// hand-overs (2 000 cycles per pass: the coarse stage's range-30 search spent 15-20 k cycles on ~25 candidates).  Integer
// arithmetic: the same numbers.
__device__ __forceinline__ int lane_zmssd_lds(const SearchWin& sw, int w, int h, int mx, int my, bool have, const unsigned Tw[16], int tsum, int tsumsq) {
    const bool inb = have && mx >= 4 && my >= 4 && mx < w - 4 && my < h - 4;
    const int off0 = inb ? (my - 4 - sw.y0) * SW_SIDE + (mx - 4 - sw.x0) : 0;
    const unsigned sh = (unsigned)(off0 & 3);
    const unsigned* p = sw.lds + (off0 >> 2);
    unsigned s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const unsigned a = p[r * (SW_SIDE / 4)], b = p[r * (SW_SIDE / 4) + 1], c = p[r * (SW_SIDE / 4) + 2];
        const unsigned lo = __builtin_amdgcn_alignbyte(b, a, sh), hi = __builtin_amdgcn_alignbyte(c, b, sh);
        s1 = __builtin_amdgcn_udot4(lo, 0x01010101u, s1, false);
        s1 = __builtin_amdgcn_udot4(hi, 0x01010101u, s1, false);
        s2 = __builtin_amdgcn_udot4(lo, lo, s2, false);
        s2 = __builtin_amdgcn_udot4(hi, hi, s2, false);
        s3 = __builtin_amdgcn_udot4(lo, Tw[2 * r], s3, false);
        s3 = __builtin_amdgcn_udot4(hi, Tw[2 * r + 1], s3, false);
    }
    return inb ? zmssd_finish(tsum, (int)s1, (int)s2, tsumsq, (int)s3) : PTAM_MAX_SSD + 1;
}
// win (nullable): SW_BYTES of LDS of the wave's own.  When the search region — the candidates' 8 x 8 windows around the circle of
// radius nRange — fits SW_SIDE x SW_SIDE, it is fetched once (16 bytes per lane, together with the row LUT) and the corner list
// of the row range is requested SCH chunks at a time: two or three dependent round trips per call instead of one per 64 corners
// and one per four candidates (the coarse stage's range-30 search scored ~25 candidates behind ~14 round trips: 18 k cycles).
#define SCH 6
__device__ __forceinline__ void wave_find_patch_coarse(const KfLevels& L, const ptam_patch_query& q, bool enabled, int T, int lane,
                                                       ptam_patch_result& res, unsigned* win = nullptr) {
    res.found = 0;
    res.best_ssd = PTAM_MAX_SSD + 1;
    res.best_x = res.best_y = -1;
    res.n_scored = 0;
    res.pad_ = 0;
    res.pos[0] = res.pos[1] = 0;
    bool search = q.level >= 0 && q.level < PTAM_LEVELS;
    if (!enabled) search = false;   // Finder.TemplateBad(): the point is dropped before the search (src/Tracker.cc:876-879)
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
    SearchWin sw{nullptr, 0, 0};
    if (search && win && 2 * (int)nRange + 8 <= SW_SIDE && w >= SW_SIDE && h >= SW_SIDE) {
        // origin: the region's corner, moved inside the image (the in-image part of the region stays covered: side <= SW_SIDE)
        sw.x0 = min(max(px - (int)nRange - 4, 0), w - SW_SIDE);
        sw.y0 = min(max(py - (int)nRange - 4, 0), h - SW_SIDE);
        typedef unsigned u32_unaligned __attribute__((aligned(1)));
        const uint8_t* src = im + (size_t)(sw.y0 + (lane >> 1)) * w + sw.x0 + 16 * (lane & 1);
        const unsigned a0 = *(const u32_unaligned*)src, a1 = *(const u32_unaligned*)(src + 4), a2 = *(const u32_unaligned*)(src + 8),
                       a3 = *(const u32_unaligned*)(src + 12);
        ((uint4*)win)[lane] = make_uint4(a0, a1, a2, a3);   // row lane / 2, bytes 16 (lane % 2) .. + 15
        if (lane < 4) win[SW_SIDE * SW_SIDE / 4 + lane] = 0;   // (the word past the last one, read by the byte shift)
        sw.lds = win;
        // (lanes read what OTHER lanes of the wave stored: in-order LDS and the compiler's ignorance of aliasing made it work;
        //  the memory model wants a fence and a wave barrier — which cost nothing at run time)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (search && sw.lds) {
        // ---- lane-per-candidate path: the passing corners of the row range are compacted, in corner order, into a list in
        // LDS; every 64 of them (and the rest at the end) are scored at once, and the first strict minimum in corner order is the
        // minimum of (score << 7 | position in the batch), earlier batches winning ties ----
        const int tsum = wave_sum_i32(T), tsumsq = wave_sum_i32(T * T);
        unsigned Tw[16];   // the template as 16 words (row r: columns 0-3, 4-7), wave-uniform
        {
            const unsigned t0 = (unsigned)__shfl(T, (lane * 4) & 63, 64), t1 = (unsigned)__shfl(T, (lane * 4 + 1) & 63, 64),
                           t2 = (unsigned)__shfl(T, (lane * 4 + 2) & 63, 64), t3 = (unsigned)__shfl(T, (lane * 4 + 3) & 63, 64);
            const unsigned packed = t0 | (t1 << 8) | (t2 << 16) | (t3 << 24);   // (lanes 0 .. 15 hold the 16 words)
#pragma unroll
            for (int k = 0; k < 16; k++) Tw[k] = (unsigned)__builtin_amdgcn_readlane((int)packed, k);
        }
        int2* list = (int2*)(win + (SW_SIDE * SW_SIDE + 16) / 4);
        int best = PTAM_MAX_SSD + 1, bx = -1, by = -1, nsc = 0, head = 0, tail = 0;   // list entries [head, tail), positions mod SW_LIST
        auto score_batch = [&](int nb) {
            const int2 e = list[(head + min(lane, nb - 1)) & (SW_LIST - 1)];
            const int ssd = lane_zmssd_lds(sw, w, h, e.x, e.y, lane < nb, Tw, tsum, tsumsq);
            const unsigned key = lane < nb ? (((unsigned)ssd << 7) | (unsigned)lane) : 0xffffffffu;
            const unsigned mk = wave_min_u32(key);
            const int mssd = (int)(mk >> 7), ml = (int)(mk & 127u);
            if (mssd < best) {
                best = mssd;
                bx = __builtin_amdgcn_readlane(e.x, ml);
                by = __builtin_amdgcn_readlane(e.y, ml);
            }
            nsc += nb;
            head += nb;
        };
        for (int base0 = i0; base0 < i1; base0 += 64 * SCH) {
            ptam_int2 cc[SCH];
#pragma unroll
            for (int u = 0; u < SCH; u++) cc[u] = corners[min(base0 + 64 * u + lane, max(i1 - 1, 0))];
#pragma unroll
            for (int u = 0; u < SCH; u++) {
                const int idx = base0 + 64 * u + lane;
                bool pass = false;
                if (idx < i1) {
                    const int dx = px - cc[u].x, dy = py - cc[u].y;
                    pass = !(cc[u].x < nLeft || cc[u].x > nRight) && !((unsigned)(dx * dx + dy * dy) > nRange * nRange);
                }
                const unsigned long long m = __ballot(pass);
                if (m) {
                    const int pos = tail + __popcll(m & ((1ull << lane) - 1ull));
                    if (pass) list[pos & (SW_LIST - 1)] = make_int2(cc[u].x, cc[u].y);
                    tail += __popcll(m);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the list is read back by other lanes of the wave)
                    __builtin_amdgcn_wave_barrier();
                    if (tail - head >= 64) score_batch(64);
                }
            }
        }
        if (tail > head) score_batch(tail - head);
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
    } else if (search) {
        const int tsum = wave_sum_i32(T), tsumsq = wave_sum_i32(T * T);
        const unsigned T4 = wave_pack_template4(T, lane);   // candidates are scored four at a time (wave_zmssd4)
        int best = PTAM_MAX_SSD + 1, bx = -1, by = -1, nsc = 0;
        for (int base0 = i0; base0 < i1; base0 += 64 * SCH) {
          // (SCH chunks of the corner list requested together)
          ptam_int2 cc[SCH];
#pragma unroll
          for (int u = 0; u < SCH; u++) {
              const int idx = base0 + 64 * u + lane;
              cc[u] = corners[min(idx, max(i1 - 1, 0))];
          }
#pragma unroll
          for (int u = 0; u < SCH; u++) {
            const int base = base0 + 64 * u;
            if (base >= i1) break;
            const int idx = base + lane;
            ptam_int2 c = {0, 0};
            bool pass = false;
            if (idx < i1) {
                c = cc[u];
                const int dx = px - c.x, dy = py - c.y;
                pass = !(c.x < nLeft || c.x > nRight) && !((unsigned)(dx * dx + dy * dy) > nRange * nRange);
            }
            unsigned long long m = __ballot(pass);
            while (m) {
                int cx[4] = {0, 0, 0, 0}, cy[4] = {0, 0, 0, 0}, n = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (m) {   // (wave-uniform)
                        const int b = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        cx[k] = __builtin_amdgcn_readlane(c.x, b);
                        cy[k] = __builtin_amdgcn_readlane(c.y, b);
                        n = k + 1;
                    }
                const int ssd_l = wave_zmssd4(im, w, h, cx, cy, n, T4, tsum, tsumsq, lane, sw);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (k < n) {   // in corner order: the first strict minimum wins
                        const int ssd = __builtin_amdgcn_readlane(ssd_l, 16 * k + 15);
                        nsc++;
                        if (ssd < best) {
                            best = ssd;
                            bx = cx[k];
                            by = cy[k];
                        }
                    }
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
}

// (reciprocals by v_rcp_f64 + two Newton steps, <= 1 ulp, as in the pose and bundle kernels: twelve IEEE divisions — a dependent
//  chain of ~12 instructions each — were most of this routine, which sits at the head of every sub-pixel refinement)
__device__ __forceinline__ void ldlt3_inverse(double A[9], double out[9]) {   // TooN Cholesky<3>::get_inverse
    double inv_d[3];
#pragma unroll
    for (int col = 0; col < 3; col++) {
        double inv_diag = 1;
#pragma unroll
        for (int row = col; row < 3; row++) {
            double val = A[row * 3 + col];
#pragma unroll
            for (int c2 = 0; c2 < col; c2++) val -= A[c2 * 3 + col] * A[row * 3 + c2];
            if (row == col) {
                A[row * 3 + col] = val;
                inv_diag = rcp_nr(val);
                inv_d[col] = inv_diag;
            } else {
                A[col * 3 + row] = val;
                A[row * 3 + col] = val * inv_diag;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double y[3], x[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double val = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < i; j++) val -= A[i * 3 + j] * y[j];
            y[i] = val;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) y[i] *= inv_d[i];
#pragma unroll
        for (int i = 2; i >= 0; i--) {
            double val = y[i];
#pragma unroll
            for (int j = i + 1; j < 3; j++) val -= A[j * 3 + i] * x[j];
            x[i] = val;
        }
#pragma unroll
        for (int r = 0; r < 3; r++) out[r * 3 + c] = x[r];
    }
}

// PatchFinder::MakeSubPixTemplate + IterateSubPixToConvergence (src/PatchFinder.cc:219-318) by one wave, T = this lane's
// template pixel
// win (nullable): 256 bytes of LDS of the wave's own — a 16 x 16 window of the level around the coarse position is fetched ONCE
// and the iterations interpolate out of it (the fit moves by fractions of a pixel per iteration: it leaves a window that
// reaches 4 pixels past the patch on every side only in pathological cases, which take the global loads as before).  Every
// iteration used to wait for a dependent global round trip — eight of them in a row on the coarse stage's critical path.
__device__ __forceinline__ void wave_subpix(const KfLevels& L, const ptam_subpix_query& q, int T, int lane, ptam_subpix_result& res,
                                            uint8_t* win = nullptr) {
    res.converged = 0;
    res.iterations = 0;
    res.pos[0] = q.coarse_pos[0];
    res.pos[1] = q.coarse_pos[1];
    res.mean_diff = 0.0;
    if (q.level >= 0 && q.level < PTAM_LEVELS) {
        const int w = L.w[q.level], h = L.h[q.level];
        const uint8_t* __restrict__ im = L.im[q.level];
        const int px = lane & 7, py = lane >> 3;
        const bool inner = px >= 1 && px <= 6 && py >= 1 && py <= 6;
        // MakeSubPixTemplate :219-240
        const int Txp = __shfl(T, (lane + 1) & 63, 64), Txm = __shfl(T, (lane - 1) & 63, 64);
        const int Typ = __shfl(T, (lane + 8) & 63, 64), Tym = __shfl(T, (lane - 8) & 63, 64);
        const double gx = inner ? 0.5 * (Txp - Txm) : 0.0, gy = inner ? 0.5 * (Typ - Tym) : 0.0;
        const double one = inner ? 1.0 : 0.0;
        double H[9];
        H[0] = wave_sum_f64(gx * gx);
        H[1] = H[3] = wave_sum_f64(gx * gy);
        H[2] = H[6] = wave_sum_f64(gx * one);
        H[4] = wave_sum_f64(gy * gy);
        H[5] = H[7] = wave_sum_f64(gy * one);
        H[8] = wave_sum_f64(one);
        double Hinv[9];
        ldlt3_inverse(H, Hinv);
        const double jx = (double)(float)gx, jy = (double)(float)gy;   // mimJacs holds floats
        double pos0 = q.coarse_pos[0], pos1 = q.coarse_pos[1], mean_diff = 0.0;
        const int scale = 1 << q.level;
        const double inv_scale = 1.0 / scale;   // (a power of two: x * inv_scale IS x / scale, without the division's twelve instructions)
        // the window: rows wy0 .. wy0 + 15, columns wx0 .. wx0 + 15 (lane = 4 consecutive bytes of a row), if it lies in the image
        int wx0 = 0, wy0 = 0;
        bool have_win = false;
        if (win) {
            const double c0x = (pos0 + 0.5) * inv_scale - 0.5, c0y = (pos1 + 0.5) * inv_scale - 0.5;
            wx0 = (int)floor(c0x) - 8;
            wy0 = (int)floor(c0y) - 8;
            have_win = wx0 >= 0 && wy0 >= 0 && wx0 + 16 <= w && wy0 + 16 <= h;
            if (have_win) {
                const uint8_t* src = im + (size_t)(wy0 + (lane >> 2)) * w + wx0 + 4 * (lane & 3);
                const unsigned b0 = src[0], b1 = src[1], b2 = src[2], b3 = src[3];
                ((unsigned*)win)[lane] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the window is read by other lanes of the wave: see wave_find_patch_coarse)
            __builtin_amdgcn_wave_barrier();
        }
        for (int it = 0; it < q.max_its; it++) {
            res.iterations = it + 1;
            // IterateSubPix :271-318
            const double cx = (pos0 + 0.5) * inv_scale - 0.5, cy = (pos1 + 0.5) * inv_scale - 0.5;   // LevelNPos
            const int rx = (int)(cx > 0.0 ? cx + 0.5 : cx - 0.5), ry = (int)(cy > 0.0 ? cy + 0.5 : cy - 0.5);   // ir_rounded
            if (!(rx >= 5 && ry >= 5 && rx < w - 5 && ry < h - 5)) break;
            const double bx = cx - 4, by = cy - 4;
            const double dX = bx - floor(bx), dY = by - floor(by);
            const float fTL = (float)((1.0 - dX) * (1.0 - dY)), fTR = (float)(dX * (1.0 - dY));
            const float fBL = (float)((1.0 - dX) * dY), fBR = (float)(dX * dY);
            const int ibx = (int)bx, iby = (int)by;   // ::ir() truncation
            double d0 = 0, d1 = 0, d2 = 0;
            if (inner) {
                float p00, p01, p10, p11;
                const int ox = ibx - wx0, oy = iby - wy0;
                if (have_win && ox >= 0 && oy >= 0 && ox + 8 <= 15 && oy + 8 <= 15) {   // (wave-uniform: rows oy + py + 1, columns ox + px + 1 stay inside)
                    const uint8_t* p = win + (oy + py) * 16 + ox + px;
                    p00 = (float)p[0], p01 = (float)p[1], p10 = (float)p[16], p11 = (float)p[17];
                } else {
                    const uint8_t* p = im + (size_t)(iby + py) * w + ibx + px;
                    p00 = (float)p[0], p01 = (float)p[1], p10 = (float)p[w], p11 = (float)p[w + 1];
                }
                const float fPixel = nc_addf(nc_addf(nc_addf(nc_mulf(fTL, p00), nc_mulf(fTR, p01)), nc_mulf(fBL, p10)), nc_mulf(fBR, p11));
                const double dDiff = (double)fPixel - (double)T + mean_diff;
                d0 = dDiff * jx;
                d1 = dDiff * jy;
                d2 = dDiff;
            }
            const double a0 = wave_sum_f64(d0), a1 = wave_sum_f64(d1), a2 = wave_sum_f64(d2);
            const double u0 = Hinv[0] * a0 + Hinv[1] * a1 + Hinv[2] * a2;
            const double u1 = Hinv[3] * a0 + Hinv[4] * a1 + Hinv[5] * a2;
            const double u2 = Hinv[6] * a0 + Hinv[7] * a1 + Hinv[8] * a2;
            pos0 -= u0 * scale;
            pos1 -= u1 * scale;
            mean_diff -= u2;
            if (u0 * u0 + u1 * u1 < 0.03 * 0.03) {
                res.converged = 1;
                break;
            }
        }
        res.pos[0] = pos0;
        res.pos[1] = pos1;
        res.mean_diff = mean_diff;
    }
}

// PatchFinder::MakeTemplateCoarseCont (src/PatchFinder.cc:98-127) by one wave: lane = output pixel (i = lane / 8 row,
// j = lane % 8 column); returns this lane's template pixel and fills r (identical in every lane).  The source position is
// NOT evaluated in closed form: the reference walks p += across / += carriage_return pixel by pixel, and the lane replays
// that exact sequence of fp64 additions (<= 70 of them) so the sampled positions are bit-identical.  Every product and sum
// goes through the nc_* primitives: no FMA contraction.
// m2 = M2Inverse(mm2WarpInverse) * LevelScale(mnSearchLevel)   (include/Tools.h:54-65; src/PatchFinder.cc:101): {m00, m01, m10, m11}
__device__ __forceinline__ void template_m2(const TemplateJob& jb, double m2[4]) {
    const double det = nc_sub(nc_mul(jb.wi[0], jb.wi[3]), nc_mul(jb.wi[2], jb.wi[1]));
    const double inv = 1.0 / det;
    const double sc = (double)(1 << jb.search_level);
    m2[0] = nc_mul(nc_mul(jb.wi[3], inv), sc), m2[3] = nc_mul(nc_mul(jb.wi[0], inv), sc);
    m2[2] = nc_mul(nc_mul(-jb.wi[2], inv), sc), m2[1] = nc_mul(nc_mul(-jb.wi[1], inv), sc);
}
__device__ __forceinline__ int wave_make_template(const TemplateJob& jb, int lane, ptam_template_result& r) {
    r.bad = 1;
    r.n_outside = 0;
    r.sum = r.sum_sq = 0;
    r.m2[0] = r.m2[1] = r.m2[2] = r.m2[3] = 0;
    if (jb.search_level < 0 || jb.im == nullptr) return 0;
    double m2_[4];
    template_m2(jb, m2_);
    const double m00 = m2_[0], m01 = m2_[1], m10 = m2_[2], m11 = m2_[3];
    // CVD::transform(in, out, M, inOrig = vec(irCenter), outOrig = (4,4))
    const int w = 8, h = 8;
    const double ax = m00, ay = m10;   // across = M.T()[0]
    const double dx = m01, dy = m11;   // down   = M.T()[1]
    const double p0x = nc_sub((double)jb.cx, nc_add(nc_mul(m00, 4.0), nc_mul(m01, 4.0)));
    const double p0y = nc_sub((double)jb.cy, nc_add(nc_mul(m10, 4.0), nc_mul(m11, 4.0)));
    double min_x = p0x, min_y = p0y, max_x = p0x, max_y = p0y;
    if (ax < 0) min_x = nc_add(min_x, nc_mul(w, ax)); else max_x = nc_add(max_x, nc_mul(w, ax));
    if (dx < 0) min_x = nc_add(min_x, nc_mul(h, dx)); else max_x = nc_add(max_x, nc_mul(h, dx));
    if (ay < 0) min_y = nc_add(min_y, nc_mul(w, ay)); else max_y = nc_add(max_y, nc_mul(w, ay));
    if (dy < 0) min_y = nc_add(min_y, nc_mul(h, dy)); else max_y = nc_add(max_y, nc_mul(h, dy));
    const double crx = nc_sub(dx, nc_mul(w, ax)), cry = nc_sub(dy, nc_mul(w, ay));   // carriage_return
    const bool all_inside = min_x >= 0 && min_y >= 0 && max_x < jb.w - 1 && max_y < jb.h - 1;
    // replay the walk up to my pixel
    const int i = lane >> 3, j = lane & 7;
    double px = p0x, py = p0y;
    for (int rr = 0; rr < 7; rr++)
        if (rr < i) {
#pragma unroll
            for (int cc = 0; cc < 8; cc++) {
                px = nc_add(px, ax);
                py = nc_add(py, ay);
            }
            px = nc_add(px, crx);
            py = nc_add(py, cry);
        }
    for (int cc = 0; cc < 7; cc++)
        if (cc < j) {
            px = nc_add(px, ax);
            py = nc_add(py, ay);
        }
    int v = 0, outside = 0;
    if (all_inside || (0 <= px && 0 <= py && px < (double)(jb.w - 1) && py < (double)(jb.h - 1))) {
        // CVD::sample
        const int lx = (int)px, ly = (int)py;
        const double x = nc_sub(px, (double)lx), y = nc_sub(py, (double)ly);
        const uint8_t* row0 = jb.im + (size_t)ly * jb.w + lx;
        const double a = row0[0], b = row0[1], c = row0[jb.w], d = row0[jb.w + 1];
        const double omx = nc_sub(1.0, x), omy = nc_sub(1.0, y);
        const double top = nc_add(nc_mul(omx, a), nc_mul(x, b));
        const double bot = nc_add(nc_mul(omx, c), nc_mul(x, d));
        const double val = nc_add(nc_mul(omy, top), nc_mul(y, bot));
        v = (int)(unsigned char)val;   // scalar_convert<byte, byte, double>: static_cast
    } else {
        outside = 1;   // defaultValue = byte()
    }
    const int n_out = wave_sum_i32(outside), s1 = wave_sum_i32(v), s2 = wave_sum_i32(v * v);
    r.n_outside = n_out;
    r.bad = n_out != 0;
    r.sum = s1;      // MakeTemplateSums src/PatchFinder.cc:326-... : sum and sum of squares of all 64 pixels
    r.sum_sq = s2;
    r.m2[0] = m00, r.m2[1] = m01, r.m2[2] = m10, r.m2[3] = m11;
    return v;
}
