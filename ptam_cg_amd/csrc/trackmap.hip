// trackmap.hip — Tracker::TrackMap (src/Tracker.cc:442-696) as ONE device-resident chain.
//
// The reference's frame is  PVS loop (:453-478)  ->  choice of the coarse / top-level / fine search sets (:480-611)  ->
// SearchForPoints on the coarse set, range 30, 8 sub-pixel iterations (:549)  ->  ten coarse pose iterations (:554-568)  ->
// SearchForPoints on the remaining top-level points (sub-pixel) and on the fine set (:572-610)  ->  ten fine pose
// iterations (:613-643)  ->  measurement / scene-depth bookkeeping (:660-696).  Which sets are searched, whether the coarse
// stage counts (nFound >= CoarseMin), the fine search range (5 or 10) and every list length are decided from data that only
// exists on the device, so the whole chain is enqueued at once: list lengths, slot ranges and the stage flags live in a
// small control block (TmCtl) that the kernels read; nothing comes back to the host before the final result.
//
// std::random_shuffle (:483-484, :598) is replaced by caller-provided permutations of the map-point ids: level l's
// list is taken in the order its members appear in `shuffle_levels` (the order a uniform permutation induces on a subset is
// itself uniform), the chop of the fine set likewise with `shuffle_fine`.
//
// Per map point the chain keeps the reference's TrackerData (v3Cam, v2Image, m2CamDerivs) in the PVS result table and
// carries it from stage to stage exactly as the reference does: the fine loop's iteration 0 does not re-project
// (:617), so a point found in the coarse stage enters it with the state the coarse loop left, a point of the later sets
// with its re-projection at the post-coarse pose and the camera derivatives of the PVS pass (ProjectAndDerivs only
// refreshes them for found points, include/Tracker.h:89-94).
#include <atomic>
#include <chrono>
#include <thread>

#include "track_internal.h"
#include "../../include/ptam_hip_bench.h"
#include "patch_device.h"
#include "keyframe_device.h"
#include "pvs_device.h"
#include "pose_device.h"

struct TmSrc {   // MapPoint::pPatchSourceKF / nSourceLevel / irCenter, resolved to the level image
    const uint8_t* im;
    int w, h;
    int cx, cy;
};

struct TmCtl {
    int n_lvl[4];           // avPVS[l].size() after the PVS loop
    int nC, nH, nF, n_slots;
    int do_coarse;          // the coarse search runs (:519)
    int did_coarse;         // mbDidCoarse (:551)
    int n_found_coarse;     // nFound of the coarse SearchForPoints
    int n_meas_coarse;      // measurements of the coarse pose loop (0 unless did_coarse)
    int n_meas;             // found entries of vIterationSet
    int fine_range;         // nFineRange (:572)
    int attempted[4], found[4];   // manMeasAttempted / manMeasFound
    int range_all[2], range_c[2], range_hf[2], range_h[2];   // slot ranges {first, end} of the stages
    int n_reused;           // searched patches whose PatchFinder kept its template (src/PatchFinder.cc:103-111)
    int outlier_by_slot;    // d.outlier is indexed by search slot (fused pose kernels) instead of by compacted measurement
    double depth[3];        // sum z, sum z^2, count over the found points (:680-690)
};

// The PatchFinder of a map point (TrackerData::Finder, include/Tracker.h:52), the part of its state that outlives a frame:
// MakeTemplateCoarseCont (src/PatchFinder.cc:98-127) keeps the template — and mbTemplateBad with it — when the finder last
// warped THIS point and neither column of the warp m2 has moved by more than 0.07 since.  (The finder is per point here, so
// "the same map point" is "a template was made at all".)
struct TmFinder {
    double m2[4];           // mm2LastWarpMatrix {m00, m01, m10, m11}
    int valid;              // mpLastTemplateMapPoint == &p
    int bad;                // mbTemplateBad
    int sum, sum_sq;        // MakeTemplateSums
    uint8_t tpl[64];        // mimTemplate
};

// What a search slot hands to the pose loop, written by the search wave when it is done with the slot, FIELD-MAJOR (plane f holds
// field f of every slot): the pose kernel's prologue needs a single round trip — status, point id, position and TrackerData state
// used to be four arrays, two of them behind the point id — and its loads are coalesced (slot-major 120-byte records cost a wave 60
// load instructions of 64 cache lines each: 8.7 k cycles of the fine loop's prologue).
//   planes 0-2 v3WorldPos | 3-4 v2Found | 5-7 v3Cam | 8-9 v2Image | 10-13 m2CamDerivs  (doubles);  tag: {status word, map point}
#define TM_REC_F 14
struct TmDev {
    int n, cap;
    ptam_pvs_point* pts;
    TmSrc* src;
    int* perm_a;
    int* perm_b;
    ptam_pvs_result* pvs;     // proj = the point's TrackerData state
    int* lvl_list;            // [4][cap]
    uint8_t* isfc;            // [cap] member of the fine candidate set
    int* list;                // [cap] search slots: coarse | top level | fine
    ptam_template_result* tres;
    ptam_patch_query* q;
    ptam_patch_result* r;
    ptam_subpix_result* sr;
    int* slot_found;
    int* slot_stat;           // per slot: found | attempted << 1 | level << 2, written by the search kernel
    int* slot_subpix;
    double2* slot_v2;
    double* recf;             // [TM_REC_F][cap] per slot: everything the pose loop wants of it (written by the search wave, state refreshed by the coarse pose loop)
    int2* rect;               // [cap] {slot_stat, map point}
    ptam_pose_meas* meas;
    ptam_projection* entry;
    int* midx;                // measurement -> map point
    int* mslot;               // measurement -> slot
    int* outlier;             // per measurement (fine loop, iteration 9)
    TmCtl* ctl;
    double* pose;
    TmFinder* finder;         // [cap] per-point PatchFinder state
};

// exclusive block scan of four counters at once (1024 threads): returns this thread's four offsets, totals in tot[]
__device__ __forceinline__ void block_scan4(const int v[4], int off[4], int tot[4], int (*wsum)[16]) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int incl[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const int x = wave_incl_scan_i32(v[l]);   // (DPP: a __shfl_up ladder is six LDS round trips per counter)
        incl[l] = x;
        if (lane == 63) wsum[l][wid] = x;
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 4; l++) {
        int base = 0, total = 0;
        for (int w = 0; w < 16; w++) {
            const int s = wsum[l][w];
            if (w < wid) base += s;
            total += s;
        }
        off[l] = base + incl[l] - v[l];
        tot[l] = total;
    }
    __syncthreads();
}

// The choice of the search sets, src/Tracker.cc:480-611.  One workgroup.  The level lists and the fine-candidate flags live
// in LDS when the map has at most TM_SEL_LDS points (global scratch otherwise): a list written to global memory and read
// back by another thread of the workgroup costs two round trips per pass, and this kernel is a chain of such passes.
#ifndef TM_SEL_LDS
#define TM_SEL_LDS 8192   // 4 x 32 KB level lists + 8 KB flags = 136 KB of the CU's 160 KB (2048 until round 2c: a 2 561-point map
                          // cost 18 us more per frame than a 1 998-point one)
#endif
__device__ __forceinline__ void tm_select_body(const TmDev& d, const ptam_trackmap_opts& o) {   // a 1024-thread workgroup
    __shared__ int wsum[4][16];
    __shared__ int ls_list[4][TM_SEL_LDS];
    __shared__ uint8_t ls_isfc[TM_SEL_LDS];
    const int tid = threadIdx.x, n = d.n, cap = d.cap;
    const bool lds = n <= TM_SEL_LDS;
    const int chunk = (n + 1023) / 1024, p0 = min(n, tid * chunk), p1 = min(n, p0 + chunk);
    // both permutations' entries of my run and the levels of the first one's points leave together
    constexpr int RUNMAX = TM_SEL_LDS / 1024;   // (longer runs — maps that do not fit the LDS lists — reload inside the loops)
    int ida[RUNMAX], idb[RUNMAX], lva[RUNMAX];
#pragma unroll
    for (int j = 0; j < RUNMAX; j++) {
        ida[j] = idb[j] = 0;
        lva[j] = -1;
        if (chunk <= RUNMAX && p0 + j < p1) {
            ida[j] = d.perm_a[p0 + j];
            idb[j] = d.perm_b[p0 + j];
        }
    }
#pragma unroll
    for (int j = 0; j < RUNMAX; j++)
        if (chunk <= RUNMAX && p0 + j < p1) lva[j] = d.pvs[ida[j]].level;
    // f(id of the first permutation, its level, id of the second permutation) over my run, in order
    auto for_run = [&](auto&& f) {
        if (chunk <= RUNMAX) {
#pragma unroll
            for (int j = 0; j < RUNMAX; j++)
                if (p0 + j < p1) f(ida[j], lva[j], idb[j]);
        } else
            for (int p = p0; p < p1; p++) {
                const int ia = d.perm_a[p];
                f(ia, (int)d.pvs[ia].level, d.perm_b[p]);
            }
    };
    auto set_fc = [&](int id, int v) { if (lds) ls_isfc[id] = (uint8_t)v; else d.isfc[id] = (uint8_t)v; };
    auto get_fc = [&](int id) { return lds ? (int)ls_isfc[id] : (int)d.isfc[id]; };
    auto LL = [&](int l, int i) -> int { return lds ? ls_list[l][i] : d.lvl_list[l * cap + i]; };
    // level lists in the order the shuffle permutation visits their members
    int cnt[4] = {0, 0, 0, 0};
    for_run([&](int id, int l, int) {
        set_fc(id, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) cnt[q] += (l == q) ? 1 : 0;
    });
    int off[4], tot[4];
    block_scan4(cnt, off, tot, wsum);
    for_run([&](int id, int l, int) {
        if (l >= 0) {
            int at = 0;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (l == q) at = off[q]++;
            if (lds)
                ls_list[l][at] = id;
            else
                d.lvl_list[l * cap + at] = id;
        }
    });
    __syncthreads();
    const int n3 = tot[3], n2 = tot[2], n1 = tot[1], n0 = tot[0];
    const int cmax = (int)o.coarse_max;
    // coarse set (:519-546)
    const int do_coarse = o.try_coarse && ((unsigned)(n3 + n2) > o.coarse_min);
    int t3 = 0, t2 = 0, nC3 = 0, nC2 = 0;
    if (do_coarse) {
        t3 = min(n3, cmax);   // all of level 3, or CoarseMax of it; either way they leave avPVS[3]
        nC3 = t3;
        if (nC3 < cmax) {
            const int need = cmax - nC3;
            if (n2 <= need) {
                // :538 ASSIGNS avPVS[LEVELS-2] to vNextToSearch: the level-3 picks made above are dropped from the frame
                // (they are neither searched nor left in the PVS).  Kept as the reference does it.
                nC3 = 0;
                nC2 = n2;
                t2 = n2;
            } else {
                nC2 = need;
                t2 = need;
            }
        }
    }
    const int nC = nC3 + nC2, nH = n3 - t3;
    const int nFC = (n2 - t2) + n1 + n0;
    const int n_use = max(0, o.max_patches - (nC + nH));   // :594-596
    const bool chop = nFC > n_use;
    const int nF = chop ? n_use : nFC;
    for (int s = tid; s < nC; s += 1024) d.list[s] = s < nC3 ? LL(3, s) : LL(2, s - nC3);
    for (int s = tid; s < nH; s += 1024) d.list[nC + s] = LL(3, t3 + s);
    // fine candidates in the order of :588-590: levels 2, 1, 0
    for (int s = tid; s < nFC; s += 1024) {
        const int a = n2 - t2;
        const int id = s < a ? LL(2, t2 + s) : (s < a + n1 ? LL(1, s - a) : LL(0, s - a - n1));
        if (chop)
            set_fc(id, 1);
        else
            d.list[nC + nH + s] = id;
    }
    if (chop) {
        // random_shuffle + resize (:597-600): the first n_use members in the order of the second permutation
        __syncthreads();
        int c4[4] = {0, 0, 0, 0};
        for_run([&](int, int, int id) { c4[0] += get_fc(id); });
        int o4[4], t4[4];
        block_scan4(c4, o4, t4, wsum);
        int k = o4[0];
        for_run([&](int, int, int id) {
            if (get_fc(id)) {
                if (k < n_use) d.list[nC + nH + k] = id;
                k++;
            }
        });
    }
    if (tid == 0) {
        TmCtl& c = *d.ctl;
        c.n_lvl[0] = n0, c.n_lvl[1] = n1, c.n_lvl[2] = n2, c.n_lvl[3] = n3;
        c.nC = nC, c.nH = nH, c.nF = nF, c.n_slots = nC + nH + nF;
        c.do_coarse = do_coarse;
        c.did_coarse = 0;
        c.n_found_coarse = c.n_meas_coarse = c.n_meas = 0;
        c.n_reused = 0;
        c.fine_range = 10;
        for (int l = 0; l < 4; l++) c.attempted[l] = c.found[l] = 0;
        c.range_all[0] = 0, c.range_all[1] = nC + nH + nF;
        c.range_c[0] = 0, c.range_c[1] = nC;
        c.range_hf[0] = nC, c.range_hf[1] = nC + nH + nF;
        c.range_h[0] = nC, c.range_h[1] = nC + nH;
        c.depth[0] = c.depth[1] = c.depth[2] = 0;
    }
}

__global__ void __launch_bounds__(1024) tm_select_kernel(TmDev d, ptam_trackmap_opts o) { tm_select_body(d, o); }

// ---- horizontal fusion for a frame that arrives with its image (ptam_track_map_frame) ----
// The PVS pass and the set choice do not look at the image, the pyramid and the corner compaction do not look at the map:
// run on separate queues the two cross-queue waits cost more than the overlap (measured, see track_map_impl), but as
// workgroups of ONE launch they overlap for free — every kernel of this chain leaves most of the chip idle.
//   launch 1: pyramid (its 2-D grid linearised) | PVS pass        launch 2: FAST detect
//   launch 3: set choice (workgroup 0) | raster-ordered corner compaction (9 independent workgroups at 640x480)
template <int VARIANT>
__global__ void __launch_bounds__(256) tm_pyr_pvs_kernel(PyrArgs a, int gx, int n_pyr, DevCam cam, int n, const ptam_pvs_point* __restrict__ pts,
                                                         ptam_pvs_result* __restrict__ out, PoseArg pv, double* __restrict__ pose_out,
                                                         int* __restrict__ finder_bad) {
    const int b = blockIdx.x;
    if (b < n_pyr)
        pyramid_body<VARIANT>(a, (b % gx) * 64 + (threadIdx.x & 63), (b / gx) * 4 + (threadIdx.x >> 6));
    else
        track_pvs_body(cam, n, pts, pose_out, out, nullptr, pv, pose_out, b - n_pyr, finder_bad, (int)sizeof(TmFinder));
}
// (round 4) launch 1 of a tracked frame: the keyframe's tiles — pyramid pixels + FAST test per tile (keyframe_device.h:
// kf_fused_tile_body) — beside the PVS pass; launch 2: set choice | corner compaction.  One launch and the pyramid kernel less.
template <int VARIANT>
__global__ void __launch_bounds__(256) tm_kf_pvs_kernel(PyrArgs a, KfLevels L, int n_tiles, DevCam cam, int n, const ptam_pvs_point* __restrict__ pts,
                                                        ptam_pvs_result* __restrict__ out, PoseArg pv, double* __restrict__ pose_out,
                                                        int* __restrict__ finder_bad) {
    // workgroup order = start order: the PVS blocks first, then the tiles from the coarsest level (the longest cascade) down
    const int n_pvs = (int)gridDim.x - n_tiles, b = blockIdx.x;
    if (b < n_pvs)
        track_pvs_body(cam, n, pts, pose_out, out, nullptr, pv, pose_out, b, finder_bad, (int)sizeof(TmFinder));
    else
        kf_fused_tile_body<VARIANT>(a, L, n_tiles - 1 - (b - n_pvs));
}
__global__ void __launch_bounds__(1024) tm_compact_select_kernel(KfLevels L, TmDev d, ptam_trackmap_opts o) {
    if (blockIdx.x == 0)
        tm_select_body(d, o);
    else
        fast_compact_body(L, blockIdx.x - 1, 0);
}


// Tracker::SearchForPoints (src/Tracker.cc:867-912) for the slots of a stage, ONE WAVE PER SLOT, everything a patch needs
// in one pass of that wave: [stage 1: TrackerData::Project at the current pose — :573-574 always for the top-level set,
// :606-608 for the fine set if the coarse stage counted; bFound is false here, so the derivatives stay, include/Tracker.h:89-94]
// -> MakeTemplateCoarseCont (:873; the template never leaves the wave's registers) -> FindPatchCoarse at ir(v2Image) (:881)
// -> MakeSubPixTemplate + IterateSubPixToConvergence where the stage asks for it (:896-906: the coarse set with
// CoarseSubPixIts, the top-level set with 8, the fine set not at all).  Every lane computes the same scalars (wave-uniform
// control flow); lane 0 stores.  stage 0: the coarse set; stage 1: top-level and fine sets.
__device__ __forceinline__ void tm_search_body(const DevCam& cam, const KfLevels& L, const TmDev& d, int stage, unsigned coarse_range,
                                               int coarse_its, int bx) {   // a 256-thread workgroup: four slots
    // (the slot is the workgroup's own — the launch covers the slots from 0 — so that the point id is requested together with
    //  the control block, not behind it: one dependent round trip less at the head of every search wave)
    const int s = bx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
#ifdef K7_TIMING
    const long long ts0_ = (long long)__builtin_readcyclecounter();
#endif
    const int id = d.list[min(s, d.cap - 1)];
    const TmCtl& c = *d.ctl;
    const int first = stage == 0 ? c.range_c[0] : c.range_hf[0], end = stage == 0 ? c.range_c[1] : c.range_hf[1];
    if (s < first || s >= end) return;
    ptam_pvs_result& pv = d.pvs[id];
    // the point's PatchFinder state, requested with the point's TrackerData (one round trip, not one more behind the decision)
    TmFinder& fs = d.finder[id];
    const double fm0 = fs.m2[0], fm1 = fs.m2[1], fm2 = fs.m2[2], fm3 = fs.m2[3];
    const int f_valid = fs.valid, f_bad = fs.bad, f_sum = fs.sum, f_sum_sq = fs.sum_sq, f_tpl = fs.tpl[lane];
    double u = pv.proj.image[0], v = pv.proj.image[1];
    const ptam_pvs_point& p = d.pts[id];
    const double pw0 = p.world[0], pw1 = p.world[1], pw2 = p.world[2];
    const TmSrc sr = d.src[id];   // (with the point's other records: behind the re-projection it was a round trip of its own)
    const int pv_level = pv.level;
    double pwi[4];
#pragma unroll
    for (int k = 0; k < 4; k++) pwi[k] = pv.warp_inverse[k];
    double pcam0 = pv.proj.cam[0], pcam1 = pv.proj.cam[1], pcam2 = pv.proj.cam[2];
    const double pd0 = pv.proj.derivs[0], pd1 = pv.proj.derivs[1], pd2 = pv.proj.derivs[2], pd3 = pv.proj.derivs[3];
    if (stage == 1 && (s < c.range_h[1] || c.did_coarse)) {
        double X, Y, Z;
        se3_apply(d.pose, pw0, pw1, pw2, X, Y, Z);
        int in_image = 0;
        bool reached = false;
        if (!(Z < 0.001)) {
            const double x = X / Z, y = Y / Z;
            if (!(x * x + y * y > cam.largest_radius * cam.largest_radius)) {
                double rr, f;
                cam_project(cam, x, y, u, v, rr, f);
                reached = true;
                if (!(rr > cam.max_r) && !(u < 0 || v < 0 || u > cam.width || v > cam.height)) in_image = 1;
            }
        }
        if (lane == 0) {
            pv.proj.cam[0] = X, pv.proj.cam[1] = Y, pv.proj.cam[2] = Z;
            if (reached) pv.proj.image[0] = u, pv.proj.image[1] = v;
            pv.proj.in_image = in_image;
        }
        pcam0 = X, pcam1 = Y, pcam2 = Z;   // (u, v: the new position if the projection got that far, the old one otherwise)
    }
    ptam_patch_query q;
    q.x = (int)u;   // ir(): truncation
    q.y = (int)v;
    q.level = pv_level;
    q.range = stage == 0 ? coarse_range : (unsigned)c.fine_range;
    TemplateJob jb;
    jb.im = sr.im;
    jb.w = sr.w;
    jb.h = sr.h;
    jb.search_level = pv_level;
    jb.cx = sr.cx;
    jb.cy = sr.cy;
#pragma unroll
    for (int k = 0; k < 4; k++) jb.wi[k] = pwi[k];
    // MakeTemplateCoarseCont (src/PatchFinder.cc:98-127): re-make the template unless this finder's last one was made with
    // (nearly) this warp — then the template, its sums and mbTemplateBad stay as they are
    ptam_template_result tr;
    int T, kept = 0;
#ifdef K7_TIMING
    const long long ts1_ = (long long)__builtin_readcyclecounter();
#endif
    {
        double m2[4];
        template_m2(jb, m2);
        const double c0x = m2[0] - fm0, c0y = m2[2] - fm2, c1x = m2[1] - fm1, c1y = m2[3] - fm3;   // columns m2.T()[0], m2.T()[1]
        const double lim = 0.07 * 0.07;
        const bool refresh = !f_valid || pv_level < 0 || c0x * c0x + c0y * c0y > lim || c1x * c1x + c1y * c1y > lim;
        if (refresh) {
            T = wave_make_template(jb, lane, tr);
            fs.tpl[lane] = (uint8_t)T;
            if (lane == 0 && pv_level >= 0) {
                fs.m2[0] = tr.m2[0], fs.m2[1] = tr.m2[1], fs.m2[2] = tr.m2[2], fs.m2[3] = tr.m2[3];
                fs.valid = 1;
                fs.bad = tr.bad;
                fs.sum = tr.sum;
                fs.sum_sq = tr.sum_sq;
            }
        } else {
            T = f_tpl;
            tr.bad = f_bad;
            tr.n_outside = 0;
            tr.sum = f_sum;
            tr.sum_sq = f_sum_sq;
            tr.m2[0] = fm0, tr.m2[1] = fm1, tr.m2[2] = fm2, tr.m2[3] = fm3;
            kept = 1;   // (counted by the gather pass from the slot's status word: a thousand atomics on one counter cost 5 us)
        }
    }
    ptam_patch_result res;
#ifdef K7_TIMING
    const long long ts2_ = (long long)__builtin_readcyclecounter();
#endif
    __shared__ __attribute__((aligned(16))) unsigned sw_win[4][SW_BYTES / 4];   // (a wave's own search region: no barrier)
    wave_find_patch_coarse(L, q, !tr.bad, T, lane, res, sw_win[threadIdx.x >> 6]);
#ifdef K7_TIMING
    const long long ts3_ = (long long)__builtin_readcyclecounter();
#endif
    if (lane == 0) {
        d.tres[s] = tr;
        d.q[s] = q;
        d.r[s] = res;
        if (tr.bad) pv.proj.in_image = 0;   // TD.bInImage = false (:877)
    }
    const int its = stage == 0 ? coarse_its : (s < c.range_h[1] ? 8 : 0);
    ptam_subpix_result sres;
    sres.converged = 0;
    sres.iterations = 0;
    sres.pos[0] = sres.pos[1] = 0;
    if (its > 0) {
        ptam_subpix_query sq;
        sq.level = (q.level >= 0 && !tr.bad && res.found) ? q.level : -1;
        sq.max_its = its;
        sq.coarse_pos[0] = res.pos[0];
        sq.coarse_pos[1] = res.pos[1];
        __shared__ uint8_t sp_win[4][256];   // (a wave's own 16 x 16 window: no barrier)
        wave_subpix(L, sq, T, lane, sres, sp_win[threadIdx.x >> 6]);
        if (lane == 0) d.sr[s] = sres;
    }
#ifdef K7_TIMING
    if (lane == 0 && (s == 3 || s == 40 || s == 500) && (clock64() & 0x3f000) == 0)
        printf("search wave stage %d slot %d level %d kept %d its %d(%d): loads %lld | template %lld | search %lld (scored %d) | sub-pixel %lld\n", stage, s, q.level,
               kept, its, sres.iterations, ts1_ - ts0_, ts2_ - ts1_, ts3_ - ts2_, res.n_scored, (long long)__builtin_readcyclecounter() - ts3_);
#endif
    // the slot's outcome (the tail of SearchForPoints, :880-906), for the gather pass: one packed word + the position
    double2 v2pub = make_double2(0, 0);
    if (lane == 0) {
        const bool att = q.level >= 0 && !tr.bad;   // manMeasAttempted (:880)
        int found = att && res.found, sub = 0;
        double2 v2 = make_double2(0, 0);
        if (found) {
            if (its > 0) {
                sub = 1;
                found = sres.converged;   // :898-904
                v2 = make_double2(sres.pos[0], sres.pos[1]);
            } else
                v2 = make_double2(res.pos[0], res.pos[1]);
        }
        d.slot_found[s] = found;
        d.slot_subpix[s] = sub;
        d.slot_v2[s] = v2;
        v2pub = v2;
        const int stat = found | ((int)att << 1) | ((q.level & 3) << 2) | (kept << 4);
        d.slot_stat[s] = stat;
        d.rect[s] = make_int2(stat, id);
    }
    {
        // the slot's record: lane f stores field f (one store instruction for the fourteen planes)
        const double2 v2r = make_double2(__shfl(v2pub.x, 0, 64), __shfl(v2pub.y, 0, 64));
        const double fld[TM_REC_F] = {pw0, pw1, pw2, v2r.x, v2r.y, pcam0, pcam1, pcam2, u, v, pd0, pd1, pd2, pd3};
        double mine = 0;
#pragma unroll
        for (int f = 0; f < TM_REC_F; f++) mine = lane == f ? fld[f] : mine;
        if (lane < TM_REC_F) d.recf[(size_t)lane * d.cap + s] = mine;
    }
}
__global__ void __launch_bounds__(256) tm_search_kernel(DevCam cam, KfLevels L, TmDev d, int stage, unsigned coarse_range, int coarse_its) {
    tm_search_body(cam, L, d, stage, coarse_range, coarse_its, blockIdx.x);
}

// The tail of SearchForPoints (:883-909) and the measurement list of the pose loop that follows.  One workgroup.
//   stage 0: status of the coarse slots, nFound, mbDidCoarse, the coarse loop's measurements;
//   stage 1: status of the other slots, then vIterationSet's found entries in order (coarse, top level, fine).
struct TmMailbox {
    ptam_trackmap_result res;
    double depth3[3];
    unsigned long long seq;
};
// the last act of a gather pass: thread 0 books the per-level counts and the stage's outcome (coarse: mbDidCoarse and the
// fine range; fine: everything of the frame's result but the pose and the depth sums)
__device__ __forceinline__ void tm_gather_finish(const TmDev& d, TmCtl& c, int stage, unsigned coarse_min, TmMailbox* mbox, int total,
                                                 int (*lsum)[16], int n_kept) {
    const int tid = threadIdx.x;
    int tot[1] = {total};
    if (tid == 0) {
        const int n_reused = c.n_reused + n_kept;   // searched patches of the stages so far whose finder kept its template
        c.n_reused = n_reused;
        int f4[4], a4[4];
#pragma unroll
        for (int l = 0; l < 4; l++) {   // (unrolled: dynamically indexed local arrays are scratch memory)
            int tf = 0, ta = 0;
            for (int w = 0; w < 16; w++) {
                tf += lsum[l][w];
                ta += lsum[4 + l][w];
            }
            f4[l] = c.found[l] + tf;
            a4[l] = c.attempted[l] + ta;
            c.found[l] = f4[l];
            c.attempted[l] = a4[l];
        }
        if (stage == 0) {
            c.n_found_coarse = tot[0];
            c.did_coarse = c.do_coarse && (unsigned)tot[0] >= coarse_min;   // :550-551
            c.n_meas_coarse = c.did_coarse ? tot[0] : 0;
            c.fine_range = c.did_coarse ? 5 : 10;                           // :572
        } else {
            c.n_meas = tot[0];
            c.outlier_by_slot = 0;
            // everything of the frame's result but the pose and the depth sums (the fine pose loop publishes those, then
            // the sequence word)
            ptam_trackmap_result& r = mbox->res;
            r.templates_reused = n_reused;
            r.pad_ = 0;
            r.did_coarse = c.did_coarse;
#pragma unroll
            for (int l = 0; l < 4; l++) {
                r.n_pvs[l] = c.n_lvl[l];
                r.attempted[l] = a4[l];
                r.found[l] = f4[l];
            }
            r.n_coarse = c.nC;
            r.n_top = c.nH;
            r.n_fine = c.nF;
            r.n_meas = tot[0];
        }
    }
}

// One WAVE per 64 slots, as many workgroups as the capacity needs (a single 1024-thread workgroup spent 12-20 us copying
// ~250-byte records per slot through ONE CU).  The search kernel has left a packed status word per slot, so every wave
// counts the found slots in front of its own by itself (at most 1024 coalesced words) — no scan across workgroups, no
// barrier; workgroup 0 also counts everything and books the stage's outcome.
#define TM_GATHER_THREADS 64
__device__ __forceinline__ void tm_gather_body(const TmDev& d, int stage, int coarse_its, unsigned coarse_min, TmMailbox* mbox, int bx) {   // one wave
    __shared__ int lsum[8][16];
    TmCtl& c = *d.ctl;
    const int lane = threadIdx.x;
    const int g_end = stage == 0 ? c.nC : c.n_slots;            // slots compacted by this stage
    const int st_first = stage == 0 ? 0 : c.nC;                 // slots whose status this stage decided
    const int s0 = bx * TM_GATHER_THREADS;
    if (s0 >= g_end && bx != 0) return;
    // my slot: everything that does not depend on another load leaves at once
    const int s = s0 + lane;
    const bool valid = s < g_end;
    const int sc = valid ? s : 0;
    const int st = valid ? d.slot_stat[sc] : 0;
    const int id = g_end > 0 ? d.list[sc] : 0;
    const double2 v2 = d.slot_v2[sc];
    // (the point and its TrackerData state are requested as soon as the id is there, found or not: behind the `found` test
    //  they would be one more round trip at the end of the kernel)
    const int idc = g_end > 0 ? id : 0;
    const double w0 = d.pts[idc].world[0], w1 = d.pts[idc].world[1], w2 = d.pts[idc].world[2];
    const ptam_projection pj = d.pvs[idc].proj;
    // found slots in front of my workgroup's first (workgroup 0: also the totals and the per-level counts of this stage)
    const int upto = bx == 0 ? g_end : s0;
    int before = 0, total = 0, lf[4] = {0, 0, 0, 0}, la[4] = {0, 0, 0, 0}, nk = 0;
    // (eight status words per lane and round, requested together: one word per round was a round trip per 64 slots — twenty
    //  in a row for the last workgroup of a thousand-slot list)
    for (int jb = 0; jb < upto; jb += 8 * TM_GATHER_THREADS) {
        int wq[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = jb + u * TM_GATHER_THREADS + lane;
            wq[u] = j < upto ? d.slot_stat[j] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = jb + u * TM_GATHER_THREADS + lane, w = wq[u];
            total += w & 1;
            if (j < s0) before += w & 1;
            if (bx == 0 && j >= st_first && j < upto) {
                const int l = (w >> 2) & 3;
                nk += (w >> 4) & 1;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    lf[q] += (l == q) & (w & 1);
                    la[q] += (l == q) & ((w >> 1) & 1);
                }
            }
        }
    }
    before = wave_sum_i32(before);
    const int found = valid ? (st & 1) : 0;
    const int k = __builtin_amdgcn_readlane(before, 63) + wave_incl_scan_i32(found) - found;
    if (found) {
        ptam_pose_meas m;
        m.world[0] = w0, m.world[1] = w1, m.world[2] = w2;
        m.found[0] = v2.x;
        m.found[1] = v2.y;
        m.sqrt_inv_noise = 1.0 / (double)(1 << ((st >> 2) & 3));   // :889
        d.meas[k] = m;
        d.entry[k] = pj;
        d.midx[k] = id;
        d.mslot[k] = s;
    }
    if (bx != 0) return;
    total = wave_sum_i32(total);
    nk = wave_sum_i32(nk);
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const int tf = wave_sum_i32(lf[l]), ta = wave_sum_i32(la[l]);
        if (lane == 63) {
            lsum[l][0] = tf;
            lsum[4 + l][0] = ta;
#pragma unroll
            for (int w = 1; w < 16; w++) lsum[l][w] = lsum[4 + l][w] = 0;
        }
    }
    __syncthreads();
    tm_gather_finish(d, c, stage, coarse_min, mbox, __builtin_amdgcn_readlane(total, 63), lsum, __builtin_amdgcn_readlane(nk, 63));
}
__global__ void __launch_bounds__(TM_GATHER_THREADS) tm_gather_kernel(TmDev d, int stage, int coarse_its, unsigned coarse_min, TmMailbox* mbox) {
    tm_gather_body(d, stage, coarse_its, coarse_min, mbox, blockIdx.x);
}

// ---- SearchForPoints' bookkeeping + the pose loop in ONE launch (round 4) -----------------------------------------------
// The gather pass above exists to hand the pose kernel a compacted measurement list.  The register-resident pose loop does not
// need one: a slot without a measurement is a lane with weight zero.  So the loop's prologue takes its measurements straight
// from the search stage's slots — thread (q, tid) owns slot tid + q * THREADS — and books the stage's outcome itself: one
// launch (and its boundary) less per stage, 2 x (5.6 us kernel + 2.3 us boundary) per frame.  Lists of more slots than the
// kernel holds (a top-level set that outgrows MaxPatchesPerFrame) are told to the host, which sends the gather pass and the
// general pose kernel instead.
template <int MPT, int THREADS>
struct TmSlotLoader {
    const TmDev& d;
    int stage;
    unsigned coarse_min;
    TmMailbox* mbox;
    unsigned long long long_seq;   // what to publish when the list is too long (0: the host knows the capacity suffices)
    int id[MPT];
    int n_found;

    template <class SH>
    __device__ __forceinline__ bool begin(SH& sh, int& n, int cap, SmallMeas (&t)[MPT]) {
        const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
        // (A) every load that hangs on no other: the control block (uniform), and my slots' status word, point id and found
        //     position — fetched for the capacity, masked by the list's length afterwards: a load behind `slot < length` would
        //     wait for the control block first, and every dependent round trip of this single-workgroup kernel is ~1 us of frame
        const TmCtl c0 = *d.ctl;
        int st[MPT];
#pragma unroll
        for (int q = 0; q < MPT; q++) {
            // the slot's record: ONE round trip, nothing behind the point id, every load coalesced (field-major planes)
            const int sc = min(tid + q * THREADS, d.cap - 1);
            const int2 tg = d.rect[sc];
            st[q] = tg.x;
            id[q] = tg.y;
            const double* rf = d.recf + sc;
            const size_t pl = (size_t)d.cap;
            t[q].world[0] = rf[0], t[q].world[1] = rf[pl], t[q].world[2] = rf[2 * pl];
            t[q].fnd[0] = rf[3 * pl], t[q].fnd[1] = rf[4 * pl];
#pragma unroll
            for (int k = 0; k < 3; k++) t[q].cam3[k] = rf[(5 + k) * pl];
            t[q].img[0] = rf[8 * pl], t[q].img[1] = rf[9 * pl];
#pragma unroll
            for (int k = 0; k < 4; k++) t[q].D[k] = rf[(10 + k) * pl];
        }
        const int g_end = stage == 0 ? c0.nC : c0.n_slots;
        const int st_first = stage == 0 ? 0 : c0.nC;
        if (g_end > cap) {
            if (tid == 0 && long_seq) *(volatile unsigned long long*)&mbox->seq = long_seq | POSE_CHAIN_LONG;
            return false;
        }
        n = g_end;
        int p_tot = 0, p_f01 = 0, p_f23 = 0, p_a01 = 0, p_a23 = 0;
#pragma unroll
        for (int q = 0; q < MPT; q++) {
            const int s = tid + q * THREADS;
            const bool valid = s < g_end;
            st[q] = valid ? st[q] : 0;
            id[q] = valid ? id[q] : 0;
            // per-level counts of the slots this stage decided (found | attempted << 1 | level << 2 | kept << 4), two 16-bit
            // fields to a word: a list holds at most 1024 slots
            const int w = st[q], l = (w >> 2) & 3, mine = valid && s >= st_first;
            t[q].sn = 1.0 / (double)(1 << l);   // :889
            t[q].listed = w & 1;
            if (!(w & 1)) {   // (no measurement in this slot: benign values)
                t[q].world[0] = t[q].world[1] = t[q].world[2] = t[q].fnd[0] = t[q].fnd[1] = t[q].sn = 0;
                t[q].cam3[0] = t[q].cam3[1] = 0;
                t[q].cam3[2] = 1;
                t[q].img[0] = t[q].img[1] = 0;
                t[q].D[0] = t[q].D[1] = t[q].D[2] = t[q].D[3] = 0;
            }
            p_tot += (w & 1) | ((mine ? (w >> 4) & 1 : 0) << 16);
            const int f = mine ? (w & 1) : 0, a = mine ? (w >> 1) & 1 : 0;
            p_f01 += (l == 0 ? f : 0) | ((l == 1 ? f : 0) << 16);
            p_f23 += (l == 2 ? f : 0) | ((l == 3 ? f : 0) << 16);
            p_a01 += (l == 0 ? a : 0) | ((l == 1 ? a : 0) << 16);
            p_a23 += (l == 2 ? a : 0) | ((l == 3 ? a : 0) << 16);
        }
        p_tot = wave_sum_i32(p_tot);
        p_f01 = wave_sum_i32(p_f01);
        p_f23 = wave_sum_i32(p_f23);
        p_a01 = wave_sum_i32(p_a01);
        p_a23 = wave_sum_i32(p_a23);
        // (the pose loop's histogram is not in use yet: five words per wave meet there)
        if (lane == 63) {
            sh.hist[wid * 8 + 0] = (unsigned)p_tot;
            sh.hist[wid * 8 + 1] = (unsigned)p_f01;
            sh.hist[wid * 8 + 2] = (unsigned)p_f23;
            sh.hist[wid * 8 + 3] = (unsigned)p_a01;
            sh.hist[wid * 8 + 4] = (unsigned)p_a23;
        }
        __syncthreads();
        unsigned tt = 0, f01 = 0, f23 = 0, a01 = 0, a23 = 0;
#pragma unroll
        for (int w = 0; w < THREADS / 64; w++) {
            tt += sh.hist[w * 8 + 0];
            f01 += sh.hist[w * 8 + 1];
            f23 += sh.hist[w * 8 + 2];
            a01 += sh.hist[w * 8 + 3];
            a23 += sh.hist[w * 8 + 4];
        }
        __syncthreads();   // (the words are read; the loop's own zeroing pass of the histogram follows)
        const int total = (int)(tt & 0xffffu), n_kept = (int)(tt >> 16);
        const int did = stage == 0 ? (c0.do_coarse && (unsigned)total >= coarse_min) : 1;
        if (tid == 0) {
            TmCtl& c = *d.ctl;
            const int lf[4] = {(int)(f01 & 0xffffu), (int)(f01 >> 16), (int)(f23 & 0xffffu), (int)(f23 >> 16)};
            const int la[4] = {(int)(a01 & 0xffffu), (int)(a01 >> 16), (int)(a23 & 0xffffu), (int)(a23 >> 16)};
            const int n_reused = c0.n_reused + n_kept;
            c.n_reused = n_reused;
            int f4[4], a4[4];
#pragma unroll
            for (int l = 0; l < 4; l++) {   // (unrolled: a dynamically indexed local array is scratch memory, and a kernel that
                                            //  needs scratch is dispatched later)
                f4[l] = c0.found[l] + lf[l];
                a4[l] = c0.attempted[l] + la[l];
                c.found[l] = f4[l];
                c.attempted[l] = a4[l];
            }
            if (stage == 0) {
                c.n_found_coarse = total;
                c.did_coarse = did;                                   // :550-551
                c.n_meas_coarse = did ? total : 0;
                c.fine_range = did ? 5 : 10;                          // :572
            } else {
                c.n_meas = total;
                c.outlier_by_slot = 1;
                // the frame's integer results: parked in LDS and written to the (host-mapped) result block at the kernel's end —
                // a store over PCIe issued here is waited for by this wave's next barrier (its release fence), i.e. by everybody
                int* bk = sh.book;   // ptam_trackmap_result from did_coarse on: did_coarse | n_pvs[4] | attempted[4] | found[4] | n_coarse n_top n_fine n_meas | depth_n | templates_reused | pad_
                bk[0] = c0.did_coarse;
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    bk[1 + l] = c0.n_lvl[l];
                    bk[5 + l] = a4[l];
                    bk[9 + l] = f4[l];
                }
                bk[13] = c0.nC;
                bk[14] = c0.nH;
                bk[15] = c0.nF;
                bk[16] = total;
                bk[17] = 0;
                bk[18] = n_reused;
                bk[19] = 0;
            }
        }
        n_found = did ? total : 0;
        if (!did) {   // the coarse stage found too little: no pose update (:550), nobody is listed
#pragma unroll
            for (int q = 0; q < MPT; q++) t[q].listed = 0;
        }
        return true;
    }
    __device__ __forceinline__ void load(int, int, int, SmallMeas&) const {}   // (begin has filled the slots)
    template <class SH>
    __device__ __forceinline__ void publish(SH& sh) const {
        static_assert(offsetof(ptam_trackmap_result, pad_) == offsetof(ptam_trackmap_result, did_coarse) + 19 * 4, "ptam_trackmap_result layout");
        if (stage == 1 && threadIdx.x < 20 && threadIdx.x != 17) (&mbox->res.did_coarse)[threadIdx.x] = sh.book[threadIdx.x];
    }
    __device__ __forceinline__ bool has_entry() const { return true; }
    __device__ __forceinline__ ptam_projection* td_target(int q, int, const PoseChainIo&) const { return &d.pvs[id[q]].proj; }
    // the slot's record follows the TrackerData the coarse loop leaves (the fine loop starts from it)
    __device__ __forceinline__ void td_also(int, int i, const SmallMeas& m) const {
        double* rf = d.recf + i;
        const size_t pl = (size_t)d.cap;
#pragma unroll
        for (int k = 0; k < 3; k++) rf[(5 + k) * pl] = m.cam3[k];
        rf[8 * pl] = m.img[0], rf[9 * pl] = m.img[1];
#pragma unroll
        for (int k = 0; k < 4; k++) rf[(10 + k) * pl] = m.D[k];
    }
    __device__ __forceinline__ int listed_total(int) const { return n_found; }
};

template <int MPT, int THREADS>
__device__ __forceinline__ void tm_pose_body(const DevCam& cam, const TmDev& d, int stage, unsigned coarse_min, TmMailbox* mbox,
                                             const ptam_gn_opts& opts, const PoseChainIo& io, double* updates, unsigned long long long_seq) {
    TmSlotLoader<MPT, THREADS> ld{d, stage, coarse_min, mbox, long_seq};
    // (no per-iteration record of the updates: nobody reads it in the chain, and a global store in front of a barrier is waited
    //  for by the barrier's release fence — ten times per loop, on the frame's critical path)
#if !defined(K7_TIMING) && !defined(TM_KEEP_UPDATES)   // (TM_KEEP_UPDATES: A/B builds)
    updates = nullptr;
#endif
    pose_gn_small_body<MPT, THREADS>(cam, THREADS * MPT, ld, d.pose, opts, stage == 1 ? d.outlier : nullptr, updates, nullptr, 0ull, nullptr, nullptr,
                                     io, 0);
}
template <int MPT, int THREADS>
__global__ void __launch_bounds__(THREADS) tm_pose_kernel(DevCam cam, TmDev d, int stage, unsigned coarse_min, TmMailbox* mbox, ptam_gn_opts opts,
                                                          PoseChainIo io, double* updates, unsigned long long long_seq) {
    tm_pose_body<MPT, THREADS>(cam, d, stage, coarse_min, mbox, opts, io, updates, long_seq);
}
typedef void (*tm_pose_fn)(DevCam, TmDev, int, unsigned, TmMailbox*, ptam_gn_opts, PoseChainIo, double*, unsigned long long);
struct TmBatchItem;
// instantiation for a list of at most cap slots (<= GS_LIMIT): one wave up to 64, one slot per thread up to GS_THREADS (512), GS_MPT (two) up to GS_LIMIT (1024)
static tm_pose_fn tm_pose_pick(int cap, int* threads) {
    if (cap <= GS_WAVE_LIMIT) {
        *threads = GS_WAVE_LIMIT;
        return tm_pose_kernel<1, GS_WAVE_LIMIT>;
    }
    *threads = GS_THREADS;
    return cap <= GS_THREADS ? tm_pose_kernel<1, GS_THREADS> : tm_pose_kernel<GS_MPT, GS_THREADS>;
}

// ---- MapMaker::ReFind_Common (src/MapMaker.cc:943-1020), batched over the map points of one keyframe ----
// stage 1: projection + visibility tests (:950-975), derivatives, CalcSearchLevelAndWarpMatrix (:979, verdict unused),
// the template job at the level its loop stopped at, and the search query ir(v2Image), range 4 (:988)
__global__ void __launch_bounds__(256) refind_prep_kernel(DevCam cam, int n, const ptam_pvs_point* __restrict__ pts,
                                                          const TmSrc* __restrict__ src, const double* __restrict__ pose,
                                                          TemplateJob* __restrict__ jobs, ptam_patch_query* __restrict__ q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = pose[k];
    const ptam_pvs_point p = pts[i];
    TemplateJob j;
    j.im = nullptr;
    j.w = j.h = 0;
    j.search_level = -1;
    j.cx = j.cy = 0;
    j.wi[0] = j.wi[1] = j.wi[2] = j.wi[3] = 0;
    ptam_patch_query qq;
    qq.x = qq.y = 0;
    qq.level = -1;
    qq.range = 4;
    double X, Y, Z;
    se3_apply(T, p.world[0], p.world[1], p.world[2], X, Y, Z);
    if (!(Z < 0.001)) {
        const double x = X / Z, y = Y / Z;
        if (!(x * x + y * y > cam.largest_radius * cam.largest_radius)) {
            double u, v, rr, f;
            cam_project(cam, x, y, u, v, rr, f);
            if (!(rr > cam.max_r) && !(u < 0 || v < 0 || u > cam.width || v > cam.height)) {
                double D[4];
                cam_derivs(cam, x, y, rr, f, D);
                double det = pvs_warp_matrix(T, X, Y, Z, D, p, j.wi);
                int l = 0;
                while (det > 3 && l < PTAM_LEVELS - 1) {
                    l++;
                    det *= 0.25;
                }
                const TmSrc sr = src[i];
                j.im = sr.im;
                j.w = sr.w;
                j.h = sr.h;
                j.cx = sr.cx;
                j.cy = sr.cy;
                j.search_level = l;
                qq.x = (int)u;   // ir(): truncation
                qq.y = (int)v;
                qq.level = l;
            }
        }
    }
    jobs[i] = j;
    q[i] = qq;
}
// stage 2: Finder.TemplateBad() (:982-986) takes the point out of the search
__global__ void __launch_bounds__(256) refind_mask_kernel(int n, const ptam_template_result* __restrict__ tres, ptam_patch_query* __restrict__ q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && q[i].level >= 0 && tres[i].bad) q[i].level = -2 - q[i].level;   // (remembers the level for the report)
}
// stage 3: the measurement (:995-1011)
__global__ void __launch_bounds__(256) refind_finish_kernel(int n, const ptam_patch_query* __restrict__ q, const ptam_patch_result* __restrict__ r,
                                                            const ptam_subpix_result* __restrict__ sr, ptam_refind_result* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    ptam_refind_result o;
    const int lv = q[i].level;
    o.found = 0;
    o.level = lv >= 0 ? lv : (lv <= -2 ? -2 - lv : -1);
    o.sub_pix = 0;
    o.never_retry = 1;
    o.root_pos[0] = o.root_pos[1] = 0;
    if (lv >= 0 && r[i].found) {
        o.found = 1;
        o.never_retry = 0;
        if (lv > 0) {   // sub-pixel position whether or not the iteration converged (:1000-1006)
            o.sub_pix = 1;
            o.root_pos[0] = sr[i].pos[0];
            o.root_pos[1] = sr[i].pos[1];
        } else {
            o.root_pos[0] = r[i].pos[0];
            o.root_pos[1] = r[i].pos[1];
        }
    }
    out[i] = o;
}

// ---- ReFind_Common over a list of (keyframe, point) pairs through ONE PatchFinder (src/MapMaker.cc:977, :1046-1082) ----
// The finder's state makes the list a sequence: whether pair i re-makes the template depends on the last pair that did.
// But only inside a RUN of consecutive pairs (of those that reach the finder) with the same map point: another point always
// re-makes it (src/PatchFinder.cc:103).  So: (1) every pair's exits, level, warp and m2 in parallel; (2) one workgroup
// compacts the reaching pairs and walks each run with one thread — which pair re-makes, which template a kept one uses,
// whether a rejected warp has raised mbTemplateBad since; (3) the re-made templates; (4) one wave per pair searches with its
// template; (5) the finder's state after the last pair.
struct RpPair {
    ptam_pvs_point pt;
    TmSrc src;
    double pose[12];
    long long id;
    int skip, kf;
};
struct RpFinder {   // the state of the reference's static PatchFinder that outlives a call
    long long pt;          // mpLastTemplateMapPoint
    double m2[4];          // mm2LastWarpMatrix {m00, m01, m10, m11}
    int valid, bad;        // mpLastTemplateMapPoint != NULL, mbTemplateBad
    uint8_t tpl[64];       // mimTemplate
};
struct RpDev {
    int n;
    const RpPair* pairs;
    const KfLevels* Ls;
    TemplateJob* jobs;
    ptam_patch_query* q;
    double* m2;            // [n][4]
    int* reach;            // the pair gets as far as the finder
    int* detbad;           // CalcSearchLevelAndWarpMatrix rejects the warp
    int* R;                // reaching pairs, in order
    int* nr;
    int* refresh;          // the finder re-makes its template for this pair
    int* srcp;             // pair whose template this pair searches with (-1: the template the finder brought along)
    int* dacc;             // a warp was rejected since that template was made (this pair's included)
    int* bad;              // Finder.TemplateBad() for this pair
    uint8_t* tm;           // [n][64]
    ptam_template_result* tres;
    RpFinder* st;
    ptam_refind_result* out;
    int* kept;
};
__global__ void __launch_bounds__(256) rp_prep_kernel(DevCam cam, RpDev d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n) return;
    const RpPair& pr = d.pairs[i];
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; k++) T[k] = pr.pose[k];
    const ptam_pvs_point p = pr.pt;
    TemplateJob j;
    j.im = nullptr;
    j.w = j.h = 0;
    j.search_level = -1;
    j.cx = j.cy = 0;
    j.wi[0] = j.wi[1] = j.wi[2] = j.wi[3] = 0;
    ptam_patch_query qq;
    qq.x = qq.y = 0;
    qq.level = -1;
    qq.range = 4;
    int reach = 0, detbad = 0;
    double m2[4] = {0, 0, 0, 0};
    double X, Y, Z;
    se3_apply(T, p.world[0], p.world[1], p.world[2], X, Y, Z);
    if (!pr.skip && !(Z < 0.001)) {                                                     // :947-955
        const double x = X / Z, y = Y / Z;
        if (!(x * x + y * y > cam.largest_radius * cam.largest_radius)) {               // :957-961
            double u, v, rr, f;
            cam_project(cam, x, y, u, v, rr, f);
            if (!(rr > cam.max_r) && !(u < 0 || v < 0 || u > cam.width || v > cam.height)) {   // :963-975
                double D[4];
                cam_derivs(cam, x, y, rr, f, D);
                double det = pvs_warp_matrix(T, X, Y, Z, D, p, j.wi);
                int l = 0;
                while (det > 3 && l < PTAM_LEVELS - 1) {
                    l++;
                    det *= 0.25;
                }
                detbad = (det > 3 || det < 0.25) ? 1 : 0;                               // src/PatchFinder.cc:78-81
                j.im = pr.src.im;
                j.w = pr.src.w;
                j.h = pr.src.h;
                j.cx = pr.src.cx;
                j.cy = pr.src.cy;
                j.search_level = l;
                template_m2(j, m2);
                qq.x = (int)u;   // ir(): truncation
                qq.y = (int)v;
                qq.level = l;
                reach = 1;
            }
        }
    }
    d.jobs[i] = j;
    d.q[i] = qq;
    d.reach[i] = reach;
    d.detbad[i] = detbad;
#pragma unroll
    for (int k = 0; k < 4; k++) d.m2[4 * i + k] = m2[k];
}
__global__ void __launch_bounds__(1024) rp_scan_kernel(RpDev d) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < d.n; b0 += 1024) {   // stable compaction of the reaching pairs
        const int i = b0 + tid;
        const int f = i < d.n ? d.reach[i] : 0;
        const int incl = wave_incl_scan_i32(f);
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int off = base_s, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wid) off += wsum[w];
            tot += wsum[w];
        }
        if (f) d.R[off + incl - 1] = i;
        __syncthreads();
        if (tid == 0) base_s += tot;
        __syncthreads();
    }
    const int nr = base_s;
    if (tid == 0) *d.nr = nr;
    __threadfence_block();
    __syncthreads();
    const RpFinder st = *d.st;
    const double lim = 0.07 * 0.07;
    for (int k = tid; k < nr; k += 1024) {
        const int i0 = d.R[k];
        const long long id = d.pairs[i0].id;
        const bool head = k == 0 ? !(st.valid && st.pt == id) : d.pairs[d.R[k - 1]].id != id;
        if (!head && k != 0) continue;   // (a run is walked by the thread of its first pair)
        int cur = head ? i0 : -1, acc = 0;
        double l0 = st.m2[0], l1 = st.m2[1], l2 = st.m2[2], l3 = st.m2[3];
        for (int kk = k; kk < nr; kk++) {
            const int i = d.R[kk];
            if (kk > k && d.pairs[i].id != id) break;
            const double* m = d.m2 + 4 * i;
            bool refresh = kk == k && head;
            if (!refresh) {
                const double ax = m[0] - l0, ay = m[2] - l2, bx = m[1] - l1, by = m[3] - l3;   // columns m2.T()[0], m2.T()[1]
                refresh = ax * ax + ay * ay > lim || bx * bx + by * by > lim;
            }
            if (refresh) {
                cur = i;
                l0 = m[0], l1 = m[1], l2 = m[2], l3 = m[3];
                acc = 0;
            } else
                acc |= d.detbad[i];
            d.refresh[i] = refresh;
            d.srcp[i] = cur;
            d.dacc[i] = acc;
        }
    }
}
__global__ void __launch_bounds__(256) rp_template_kernel(RpDev d) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= d.n || !d.reach[i] || !d.refresh[i]) return;
    ptam_template_result tr;
    const int T = wave_make_template(d.jobs[i], lane, tr);
    d.tm[(size_t)i * 64 + lane] = (uint8_t)T;
    if (lane == 0) d.tres[i] = tr;
}
__global__ void __launch_bounds__(256) rp_search_kernel(RpDev d) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= d.n) return;
    ptam_refind_result o;
    o.found = 0;
    o.level = -1;
    o.sub_pix = 0;
    o.never_retry = d.pairs[i].skip ? 0 : 1;
    o.root_pos[0] = o.root_pos[1] = 0;
    int kept = 0;
    if (d.reach[i]) {
        const int src = d.srcp[i], refresh = d.refresh[i];
        const int T = src >= 0 ? d.tm[(size_t)src * 64 + lane] : d.st->tpl[lane];
        const int bad = refresh ? d.tres[i].bad : (((src >= 0 ? d.tres[src].bad : d.st->bad) != 0) || d.dacc[i]);
        kept = !refresh;
        const ptam_patch_query q = d.q[i];
        o.level = q.level;
        if (lane == 0) d.bad[i] = bad;
        const KfLevels& L = d.Ls[d.pairs[i].kf];
        ptam_patch_result res;
        __shared__ __attribute__((aligned(16))) unsigned rp_win[4][SW_BYTES / 4];
        wave_find_patch_coarse(L, q, !bad, T, lane, res, rp_win[threadIdx.x >> 6]);   // :982-988 (range 4)
        if (!bad && res.found) {
            o.found = 1;
            o.never_retry = 0;
            if (q.level > 0) {                                                          // :1000-1006 (convergence is not looked at)
                ptam_subpix_query sq;
                sq.level = q.level;
                sq.max_its = 8;
                sq.coarse_pos[0] = res.pos[0];
                sq.coarse_pos[1] = res.pos[1];
                ptam_subpix_result sres;
                wave_subpix(L, sq, T, lane, sres, (uint8_t*)rp_win[threadIdx.x >> 6]);   // (the search is done with the buffer)
                o.sub_pix = 1;
                o.root_pos[0] = sres.pos[0];
                o.root_pos[1] = sres.pos[1];
            } else {
                o.root_pos[0] = res.pos[0];
                o.root_pos[1] = res.pos[1];
            }
        }
    }
    if (lane == 0) {
        d.out[i] = o;
        d.kept[i] = kept;
    }
}
__global__ void __launch_bounds__(64) rp_state_kernel(RpDev d) {
    const int nr = *d.nr, lane = threadIdx.x;
    if (nr == 0) return;
    const int last = d.R[nr - 1], src = d.srcp[last];
    if (src >= 0) {
        d.st->tpl[lane] = d.tm[(size_t)src * 64 + lane];
        if (lane == 0) {
            d.st->pt = d.pairs[last].id;
            for (int k = 0; k < 4; k++) d.st->m2[k] = d.m2[4 * src + k];
            d.st->valid = 1;
        }
    }
    if (lane == 0) d.st->bad = d.bad[last];
}

// =================================================================================================
// ---- a batch of frames in ONE chain of launches (ptam_track_map_frames_batch) ----
// A process gets four hardware queues, and a tracked frame occupies its queue for the whole dependent chain (two
// single-workgroup pose loops are 94 of its ~157 us): whatever the chip has idle, at most four frames of independent
// trackers are in flight (20 k frames/s, docs/LOG_r01_r04.md section 5).  A batch runs the SAME chain once for nb trackers: every
// launch gets a second grid dimension, workgroup row y works on frame y with that frame's arguments taken from a device
// array — the kernels' bodies are the single-frame ones.
struct TmBatchItem {
    PyrArgs pa;
    int n_pyr, n;          // pyramid workgroups (gx * gy), map points
    KfLevels L;
    TmDev d;
    PoseArg pv;
    TmMailbox* mbox;
};
template <int VARIANT>
__global__ void __launch_bounds__(256) tm_pyr_pvs_batch_kernel(const TmBatchItem* __restrict__ items, int gx, DevCam cam) {
    const TmBatchItem& it = items[blockIdx.y];
    const int b = blockIdx.x;
    if (b < it.n_pyr)
        pyramid_body<VARIANT>(it.pa, (b % gx) * 64 + (threadIdx.x & 63), (b / gx) * 4 + (threadIdx.x >> 6));
    else if ((b - it.n_pyr) * 256 < max(it.n, 1))
        track_pvs_body(cam, it.n, it.d.pts, it.d.pose, it.d.pvs, nullptr, it.pv, it.d.pose, b - it.n_pyr, &it.d.finder->bad, (int)sizeof(TmFinder));
}
__global__ void __launch_bounds__(1024) tm_compact_select_batch_kernel(const TmBatchItem* __restrict__ items, ptam_trackmap_opts o) {
    const TmBatchItem& it = items[blockIdx.y];
    if (blockIdx.x == 0)
        tm_select_body(it.d, o);
    else
        fast_compact_body(it.L, blockIdx.x - 1, 0);
}
__global__ void __launch_bounds__(256) tm_search_batch_kernel(DevCam cam, const TmBatchItem* __restrict__ items, int stage, unsigned coarse_range,
                                                              int coarse_its) {
    const TmBatchItem& it = items[blockIdx.y];
    tm_search_body(cam, it.L, it.d, stage, coarse_range, coarse_its, blockIdx.x);
}
__global__ void __launch_bounds__(TM_GATHER_THREADS) tm_gather_batch_kernel(const TmBatchItem* __restrict__ items, int stage, int coarse_its,
                                                                            unsigned coarse_min) {
    const TmBatchItem& it = items[blockIdx.y];
    tm_gather_body(it.d, stage, coarse_its, coarse_min, it.mbox, blockIdx.x);
}
// the fused bookkeeping + pose loop (tm_pose_body) for the frames of a batch: one workgroup per frame; io and scratch of the frame's
// stage come from the pose items the unfused path uses
template <int MPT, int THREADS>
__global__ void __launch_bounds__(THREADS) tm_pose_batch_kernel(DevCam cam, const TmBatchItem* __restrict__ items, const PoseBatchItem* __restrict__ pitems,
                                                                int stage, unsigned coarse_min, ptam_gn_opts opts) {
    const TmBatchItem& it = items[blockIdx.x];
    const PoseBatchItem& pi = pitems[blockIdx.x];
    tm_pose_body<MPT, THREADS>(cam, it.d, stage, coarse_min, it.mbox, opts, pi.io, pi.updates, 0ull);
}
static void tm_pose_launch_batch(ptam_ctx* ctx, int nb, int cap, const TmBatchItem* d_it, const PoseBatchItem* d_p, int stage, unsigned coarse_min,
                                 const ptam_gn_opts& g) {
    if (cap <= GS_WAVE_LIMIT)
        hipLaunchKernelGGL((tm_pose_batch_kernel<1, GS_WAVE_LIMIT>), dim3(nb), dim3(GS_WAVE_LIMIT), 0, ctx->stream, ctx->cam, d_it, d_p, stage, coarse_min, g);
    else if (cap <= GS_THREADS)
        hipLaunchKernelGGL((tm_pose_batch_kernel<1, GS_THREADS>), dim3(nb), dim3(GS_THREADS), 0, ctx->stream, ctx->cam, d_it, d_p, stage, coarse_min, g);
    else
        hipLaunchKernelGGL((tm_pose_batch_kernel<GS_MPT, GS_THREADS>), dim3(nb), dim3(GS_THREADS), 0, ctx->stream, ctx->cam, d_it, d_p, stage, coarse_min, g);
}

struct ptam_tracker {
    ptam_ctx* ctx;
    TmDev d;
    void* block;
    TmMailbox* mbox;       // host-mapped
    TmMailbox* mbox_dev;
    unsigned long long seq;
    // the frame's two permutations in host-mapped memory (maps of at most TM_SEL_LDS = 8192 points: the set-choice kernel reads
    // every entry exactly once, at its start) — no upload, i.e. two copy kernels and their boundaries less per frame (~10 us);
    // d.perm_a / d.perm_b point either here or at the device arrays
    int* perm_host;        // [2][cap], host address
    int* perm_host_dev;    // device address of the same memory
    int *perm_dev_a, *perm_dev_b;
    // argument arrays of the batches this tracker leads (ptam_track_map_frames_batch): grown on demand
    void* batch_dev;
    size_t batch_cap;
    hipStream_t last_stream;   // the queue that last ran this tracker (its own context's, or a batch's lead tracker's)
    // stage timing of ptam_track_map_frame (ptam_tracker_set_profiling): an event after every launch of the frame
    bool prof;
    hipEvent_t ev[PTAM_TS_COUNT + 1];
    double stage_ms[PTAM_TS_COUNT];
    int stage_frames;
};

extern "C" {

void ptam_trackmap_opts_default(ptam_trackmap_opts* o) {
    if (!o) return;
    o->try_coarse = 1;
    o->coarse_min = 20;          // Tracker.CoarseMin          src/Tracker.cc:492
    o->coarse_max = 60;          // Tracker.CoarseMax          :493
    o->coarse_range = 30;        // Tracker.CoarseRange        :494
    o->coarse_subpix_its = 8;    // Tracker.CoarseSubPixIts    :495
    o->max_patches = 1000;       // Tracker.MaxPatchesPerFrame :593
    o->estimator = PTAM_EST_TUKEY;
    o->pad_ = 0;
}

int ptam_tracker_create(ptam_ctx* ctx, int max_points, ptam_tracker** out) {
    ARG_TRY(ctx && out && max_points >= 1);
    HIP_TRY(hipSetDevice(ctx->device));
    ptam_tracker* t = new ptam_tracker();
    std::memset(t, 0, sizeof *t);
    t->ctx = ctx;
    const size_t cap = (size_t)max_points;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t o_pts = take(cap * sizeof(ptam_pvs_point)), o_src = take(cap * sizeof(TmSrc)), o_pa = take(cap * 4), o_pb = take(cap * 4),
                 o_pvs = take(cap * sizeof(ptam_pvs_result)), o_ll = take(cap * 16), o_fc = take(cap), o_list = take(cap * 4),
                 o_tr = take(cap * sizeof(ptam_template_result)),
                 o_q = take(cap * sizeof(ptam_patch_query)), o_r = take(cap * sizeof(ptam_patch_result)),
                 o_sr = take(cap * sizeof(ptam_subpix_result)), o_sf = take(cap * 4), o_st = take(cap * 4), o_ss = take(cap * 4), o_sv = take(cap * 16), o_rec = take(cap * TM_REC_F * 8), o_rect = take(cap * 8),
                 o_me = take(cap * sizeof(ptam_pose_meas)), o_en = take(cap * sizeof(ptam_projection)), o_mi = take(cap * 4),
                 o_ms = take(cap * 4), o_ou = take(cap * 4), o_ctl = take(sizeof(TmCtl)), o_pose = take(96),
                 o_fs = take(cap * sizeof(TmFinder));
    // every failure past this point leaves through ONE path that releases what exists so far
    auto fail = [&](const char* what) {
        ptam_set_error("ptam_tracker_create: %s failed", what);
        if (t->perm_host) hipHostFree(t->perm_host);
        if (t->mbox) hipHostFree(t->mbox);
        if (t->block) hipFree(t->block);
        delete t;
        return PTAM_E_HIP;
    };
    if (hipMalloc(&t->block, off) != hipSuccess) return fail("hipMalloc");
    (void)hipMemsetAsync(t->block, 0, off, ctx->stream);
    char* b = (char*)t->block;
    TmDev& d = t->d;
    d.n = 0;
    d.cap = max_points;
    d.pts = (ptam_pvs_point*)(b + o_pts);
    d.src = (TmSrc*)(b + o_src);
    d.perm_a = (int*)(b + o_pa);
    d.perm_b = (int*)(b + o_pb);
    d.pvs = (ptam_pvs_result*)(b + o_pvs);
    d.lvl_list = (int*)(b + o_ll);
    d.isfc = (uint8_t*)(b + o_fc);
    d.list = (int*)(b + o_list);
    d.tres = (ptam_template_result*)(b + o_tr);
    d.q = (ptam_patch_query*)(b + o_q);
    d.r = (ptam_patch_result*)(b + o_r);
    d.sr = (ptam_subpix_result*)(b + o_sr);
    d.slot_found = (int*)(b + o_sf);
    d.slot_stat = (int*)(b + o_st);
    d.slot_subpix = (int*)(b + o_ss);
    d.slot_v2 = (double2*)(b + o_sv);
    d.recf = (double*)(b + o_rec);
    d.rect = (int2*)(b + o_rect);
    d.meas = (ptam_pose_meas*)(b + o_me);
    d.entry = (ptam_projection*)(b + o_en);
    d.midx = (int*)(b + o_mi);
    d.mslot = (int*)(b + o_ms);
    d.outlier = (int*)(b + o_ou);
    d.ctl = (TmCtl*)(b + o_ctl);
    d.pose = (double*)(b + o_pose);
    d.finder = (TmFinder*)(b + o_fs);   // (cleared with the block: no finder has made a template yet)
    void* h = nullptr;
    if (hipHostMalloc(&h, sizeof(TmMailbox), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return fail("hipHostMalloc");
    t->mbox = (TmMailbox*)h;
    std::memset(h, 0, sizeof(TmMailbox));
    void* dv = nullptr;
    if (hipHostGetDevicePointer(&dv, h, 0) != hipSuccess) return fail("hipHostGetDevicePointer");
    t->mbox_dev = (TmMailbox*)dv;
    t->perm_dev_a = d.perm_a;
    t->perm_dev_b = d.perm_b;
    {
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, cap * 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return fail("hipHostMalloc");
        t->perm_host = (int*)hp;
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) return fail("hipHostGetDevicePointer");
        t->perm_host_dev = (int*)dp;
    }
    // identity shuffles until the caller sets its own
    std::vector<int> idp((size_t)max_points);
    for (int i = 0; i < max_points; i++) idp[(size_t)i] = i;
    if (hipMemcpy(d.perm_a, idp.data(), cap * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d.perm_b, idp.data(), cap * 4, hipMemcpyHostToDevice) != hipSuccess)
        return fail("hipMemcpy");
    if (ptam_stream_wait(ctx->stream) != hipSuccess) return fail("stream wait");
    *out = t;
    return PTAM_OK;
}

int ptam_tracker_destroy(ptam_tracker* t) {
    if (!t) return PTAM_OK;
    hipSetDevice(t->ctx->device);
    ptam_stream_wait(t->ctx->stream);
    if (t->block) hipFree(t->block);
    if (t->mbox) hipHostFree(t->mbox);
    if (t->perm_host) hipHostFree(t->perm_host);
    if (t->batch_dev) hipFree(t->batch_dev);
    if (t->ev[0])
        for (int i = 0; i <= PTAM_TS_COUNT; i++) hipEventDestroy(t->ev[i]);
    delete t;
    return PTAM_OK;
}

// new[i] = the finder of the point that was at prev[i] in the old map (a fresh one for prev[i] < 0 and past the new map's end)
__global__ void __launch_bounds__(256) tm_finder_carry_kernel(TmFinder* __restrict__ fresh, const TmFinder* __restrict__ old, const int* __restrict__ prev,
                                                              int n, int cap) {
    // (a finder is 112 bytes = 28 words: thread = (point, word))
    constexpr int W = (int)(sizeof(TmFinder) / 4);
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)cap * W) return;
    const int i = (int)(g / W), w = (int)(g % W);
    const int src = i < n ? prev[i] : -1;
    ((unsigned*)fresh)[g] = src >= 0 ? ((const unsigned*)old)[(long long)src * W + w] : 0u;
}
static_assert(sizeof(TmFinder) % 4 == 0, "TmFinder is copied word by word");

static int tracker_set_map_impl(ptam_tracker* t, int n, const ptam_pvs_point* pts, const ptam_template_query* src, const int32_t* prev_index) {
    ARG_TRY(t && n >= 0 && n <= t->d.cap && (n == 0 || (pts && src)));
    ptam_ctx* ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (prev_index) {   // every old point carries its finder to at most one new point
        std::vector<uint8_t> seen((size_t)std::max(t->d.n, 1), 0);
        for (int i = 0; i < n; i++) {
            ARG_TRY(prev_index[i] >= -1 && prev_index[i] < t->d.n);
            if (prev_index[i] >= 0) {
                ARG_TRY(!seen[(size_t)prev_index[i]]);
                seen[(size_t)prev_index[i]] = 1;
            }
        }
    }
    std::vector<TmSrc> s((size_t)n);
    for (int i = 0; i < n; i++) {
        const ptam_template_query& q = src[i];
        ARG_TRY(q.src_kf && q.src_level >= 0 && q.src_level < PTAM_LEVELS && q.src_kf->device == ctx->device);
        s[(size_t)i].im = q.src_kf->L.im[q.src_level];
        s[(size_t)i].w = q.src_kf->L.w[q.src_level];
        s[(size_t)i].h = q.src_kf->L.h[q.src_level];
        s[(size_t)i].cx = q.center_x;
        s[(size_t)i].cy = q.center_y;
    }
    HIP_TRY(ptam_stream_wait(ctx->stream));
    if (n > 0) {
        HIP_TRY(hipMemcpy(t->d.pts, pts, (size_t)n * sizeof(ptam_pvs_point), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->d.src, s.data(), (size_t)n * sizeof(TmSrc), hipMemcpyHostToDevice));
    }
    if (prev_index && n > 0) {
        // the map changed, its points did not: a point that was in the old map keeps its TrackerData — the PatchFinder with its
        // template, warp matrix and mbTemplateBad (include/Tracker.h:42-67 lives as long as the MapPoint)
        const size_t bf = (size_t)t->d.cap * sizeof(TmFinder);
        const size_t bfa = (bf + 255) & ~(size_t)255;
        void* s_;
        const int rc = ctx_scratch(ctx, bfa + (size_t)n * 4 + 256, &s_);   // (the queue is idle: the scratch is nobody's)
        if (rc) return rc;
        int* d_prev = (int*)((char*)s_ + bfa);
        HIP_TRY(hipMemcpy(s_, t->d.finder, bf, hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(d_prev, prev_index, (size_t)n * 4, hipMemcpyHostToDevice));
        const long long words = (long long)t->d.cap * (long long)(sizeof(TmFinder) / 4);
        hipLaunchKernelGGL(tm_finder_carry_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, t->d.finder, (const TmFinder*)s_,
                           (const int*)d_prev, n, t->d.cap);
        HIP_TRY(hipGetLastError());
        HIP_TRY(ptam_stream_wait(ctx->stream));
    } else {
        // a new map: new TrackerData, new PatchFinders (include/Tracker.h:42-67) — no template has been made, none is bad
        HIP_TRY(hipMemset(t->d.finder, 0, (size_t)t->d.cap * sizeof(TmFinder)));
    }
    if (n != t->d.n) {   // the shuffles are permutations of 0..n-1: back to the identity until the caller sets them again
        std::vector<int> idp((size_t)std::max(n, 1));
        for (int i = 0; i < n; i++) idp[(size_t)i] = i;
        t->d.perm_a = t->perm_dev_a;
        t->d.perm_b = t->perm_dev_b;
        if (n > 0) {
            HIP_TRY(hipMemcpy(t->d.perm_a, idp.data(), (size_t)n * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(t->d.perm_b, idp.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        }
    }
    t->d.n = n;
    return PTAM_OK;
}

int ptam_tracker_set_map(ptam_tracker* t, int n, const ptam_pvs_point* pts, const ptam_template_query* src) {
    return tracker_set_map_impl(t, n, pts, src, nullptr);
}
int ptam_tracker_update_map(ptam_tracker* t, int n, const ptam_pvs_point* pts, const ptam_template_query* src, const int32_t* prev_index) {
    ARG_TRY(prev_index || n == 0);
    return tracker_set_map_impl(t, n, pts, src, prev_index);
}

int ptam_tracker_set_shuffle(ptam_tracker* t, const int32_t* shuffle_levels, const int32_t* shuffle_fine) {
    ARG_TRY(t && shuffle_levels && shuffle_fine);
    const int n = t->d.n;
    if (n == 0) return PTAM_OK;
    // permutations of 0..n-1 (anything else would index outside the map on the device)
    std::vector<uint8_t> seen((size_t)n);
    for (int pass = 0; pass < 2; pass++) {
        const int32_t* p = pass ? shuffle_fine : shuffle_levels;
        std::fill(seen.begin(), seen.end(), 0);
        for (int i = 0; i < n; i++) {
            ARG_TRY(p[i] >= 0 && p[i] < n && !seen[(size_t)p[i]]);
            seen[(size_t)p[i]] = 1;
        }
    }
    HIP_TRY(hipSetDevice(t->ctx->device));
    if (n <= TM_SEL_LDS) {
        // host-mapped: the kernel of the coming frame reads the entries straight from here.  The previous frame's set choice
        // is certainly done once that frame's result has arrived; otherwise wait for the queue.
        // (a batch runs the tracker on the LEAD tracker's queue: that is the one to wait for)
        if (t->mbox->seq != t->seq) HIP_TRY(ptam_stream_wait(t->last_stream ? t->last_stream : t->ctx->stream));
        std::memcpy(t->perm_host, shuffle_levels, (size_t)n * 4);
        std::memcpy(t->perm_host + t->d.cap, shuffle_fine, (size_t)n * 4);
        t->d.perm_a = t->perm_host_dev;
        t->d.perm_b = t->perm_host_dev + t->d.cap;
        return PTAM_OK;
    }
    t->d.perm_a = t->perm_dev_a;
    t->d.perm_b = t->perm_dev_b;
    void* pin;
    int rc = ctx_pinned(t->ctx, (size_t)n * 8 + 64, &pin);
    if (rc) return rc;
    HIP_TRY(ptam_stream_wait(t->ctx->stream));   // (shared staging buffer; the previous frame has been read by now anyway)
    if (t->last_stream && t->last_stream != t->ctx->stream) HIP_TRY(ptam_stream_wait(t->last_stream));
    std::memcpy(pin, shuffle_levels, (size_t)n * 4);
    std::memcpy((char*)pin + (size_t)n * 4, shuffle_fine, (size_t)n * 4);
    HIP_TRY(hipMemcpyAsync(t->d.perm_a, pin, (size_t)n * 4, hipMemcpyHostToDevice, t->ctx->stream));
    HIP_TRY(hipMemcpyAsync(t->d.perm_b, (char*)pin + (size_t)n * 4, (size_t)n * 4, hipMemcpyHostToDevice, t->ctx->stream));
    return PTAM_OK;
}

// d_new_frame (nullable): the current keyframe is made first — KeyFrame::MakeKeyFrame_Lite of this device-resident image, in
// the same queue.  (Measured and dropped: the keyframe kernels on a second queue beside the PVS pass and the set choice, which
// do not look at the image — the two cross-queue event waits cost more than the 13 us of overlap: 207-218 vs 193-200 us.)
static int track_map_impl(ptam_tracker* t, ptam_kf* cur, const uint8_t* d_new_frame, const double pose_in[12],
                          const ptam_trackmap_opts* opts, ptam_trackmap_result* out) {
    ARG_TRY(t && cur && pose_in && out);
    ptam_ctx* ctx = t->ctx;
    ARG_TRY(cur->device == ctx->device);
    ptam_trackmap_opts o;
    if (opts)
        o = *opts;
    else
        ptam_trackmap_opts_default(&o);
    ARG_TRY(o.max_patches >= 0 && o.coarse_subpix_its >= 0 && o.coarse_subpix_its <= 64);
    // (coarse_max / coarse_min become ints on the device and size the coarse search grid: a value past INT_MAX / 2 turned the
    //  set sizes negative and the set choice wrote outside its lists; the range is a pixel radius inside a 640-pixel image)
    ARG_TRY(o.coarse_max <= (unsigned)INT_MAX / 2 && o.coarse_min <= (unsigned)INT_MAX / 2 && o.coarse_range <= 4096u);
    ARG_TRY(o.estimator == PTAM_EST_TUKEY || o.estimator == PTAM_EST_CAUCHY || o.estimator == PTAM_EST_HUBER);
    HIP_TRY(hipSetDevice(ctx->device));
    const TmDev& d = t->d;
    const int n = d.n;
    hipStream_t st = ctx->stream;
    int rc = PTAM_OK;
    const bool prof = t->prof && d_new_frame;
    if (prof) hipEventRecord(t->ev[PTAM_TS_PYR_PVS], st);
    if (d_new_frame) {
        // KeyFrame::MakeKeyFrame_Lite (src/KeyFrame.cc:18-54) of the new image, the PVS pass (:453-478, the pose rides in as an
        // argument) and the set choice (:480-611) in three launches
        PyrArgs pa;
        int gx, gy;
        kf_lite_begin(cur, d_new_frame, &pa, &gx, &gy);
        PoseArg pv{};
        std::memcpy(pv.v, pose_in, 96);
        pv.use = 1;
        const int n_pyr = gx * gy, n_pvs = std::max(1, (n + 255) / 256);
        static const bool kf_two = ptam_ab_env("PTAM_TM_KF_TWO_LAUNCHES") != nullptr;   // (A/B builds: pyramid | FAST as two launches)
        if (!kf_two) {
            const int n_tiles = cur->n_blocks;
            if (ctx->halfsample == PTAM_HALFSAMPLE_T)
                hipLaunchKernelGGL(tm_kf_pvs_kernel<PTAM_HALFSAMPLE_T>, dim3(n_tiles + n_pvs), dim3(256), 0, st, pa, cur->L, n_tiles, ctx->cam, std::max(n, 0),
                                   (const ptam_pvs_point*)d.pts, d.pvs, pv, d.pose, &d.finder->bad);
            else
                hipLaunchKernelGGL(tm_kf_pvs_kernel<PTAM_HALFSAMPLE_R>, dim3(n_tiles + n_pvs), dim3(256), 0, st, pa, cur->L, n_tiles, ctx->cam, std::max(n, 0),
                                   (const ptam_pvs_point*)d.pts, d.pvs, pv, d.pose, &d.finder->bad);
            if (prof) hipEventRecord(t->ev[PTAM_TS_DETECT], st);
        } else {
            if (ctx->halfsample == PTAM_HALFSAMPLE_T)
                hipLaunchKernelGGL(tm_pyr_pvs_kernel<PTAM_HALFSAMPLE_T>, dim3(n_pyr + n_pvs), dim3(256), 0, st, pa, gx, n_pyr, ctx->cam, std::max(n, 0),
                                   (const ptam_pvs_point*)d.pts, d.pvs, pv, d.pose, &d.finder->bad);
            else
                hipLaunchKernelGGL(tm_pyr_pvs_kernel<PTAM_HALFSAMPLE_R>, dim3(n_pyr + n_pvs), dim3(256), 0, st, pa, gx, n_pyr, ctx->cam, std::max(n, 0),
                                   (const ptam_pvs_point*)d.pts, d.pvs, pv, d.pose, &d.finder->bad);
            if (prof) hipEventRecord(t->ev[PTAM_TS_DETECT], st);
            kf_launch_detect(cur, st);
        }
        if (prof) hipEventRecord(t->ev[PTAM_TS_COMPACT_SELECT], st);
        hipLaunchKernelGGL(tm_compact_select_kernel, dim3(1 + fast_compact_blocks(cur->L)), dim3(1024), 0, st, cur->L, d, o);
    } else {
        rc = pvs_launch_dev(ctx, n, d.pts, d.pose, pose_in, d.pvs, &d.finder->bad, (int)sizeof(TmFinder));                         // :453-478 (the pose rides in as an argument)
        if (rc) return rc;
        hipLaunchKernelGGL(tm_select_kernel, dim3(1), dim3(1024), 0, st, d, o);             // :480-611
    }
    // ---- coarse stage :519-569 ----
    const int ncc = std::max(1, std::min(n, (int)o.coarse_max));
    if (prof) hipEventRecord(t->ev[PTAM_TS_SEARCH_COARSE], st);
    hipLaunchKernelGGL(tm_search_kernel, dim3((ncc + 3) / 4), dim3(256), 0, st, ctx->cam, cur->L, d, 0, o.coarse_range, o.coarse_subpix_its);
    if (prof) hipEventRecord(t->ev[PTAM_TS_GATHER_COARSE], st);
    static const bool no_fuse = ptam_ab_env("PTAM_TM_NO_FUSE") != nullptr;   // (A/B runs: the gather pass + pose kernel of rounds 2-3)
    double* d_upd;
    {
        void* s_;
        rc = pose_chain_scratch(ctx, std::max(n, 1), &s_, &d_upd);
        if (rc) return rc;
    }
    {
        ptam_gn_opts g;
        ptam_gn_opts_default(&g);
        g.nonlinear_mask = 0x3ff;       // every coarse iteration re-projects (:556-562)
        g.override_sigma_sq = 1.0;      // :565
        g.mark_outliers_iter = -1;      // CalcPoseUpdate(vIterationSet, dOverrideSigma): bMarkOutliers defaults to false
        g.estimator = o.estimator;
        PoseChainIo io{};
        io.td_base = &d.pvs[0].proj;
        io.td_index = d.midx;
        io.td_stride = (int)sizeof(ptam_pvs_result);
        if (ncc <= GS_LIMIT && !no_fuse) {
            // SearchForPoints' bookkeeping and the ten coarse iterations in one launch (the coarse set holds at most CoarseMax slots)
            if (prof) hipEventRecord(t->ev[PTAM_TS_POSE_COARSE], st);
            int thr;
            const tm_pose_fn fn = tm_pose_pick(ncc, &thr);
            hipLaunchKernelGGL(fn, dim3(1), dim3(thr), 0, st, ctx->cam, d, 0, o.coarse_min, t->mbox_dev, g, io, d_upd, 0ull);
        } else {
            hipLaunchKernelGGL(tm_gather_kernel, dim3(std::max(1, (ncc + TM_GATHER_THREADS - 1) / TM_GATHER_THREADS)), dim3(TM_GATHER_THREADS), 0, st, d, 0, o.coarse_subpix_its, o.coarse_min, t->mbox_dev);
            if (prof) hipEventRecord(t->ev[PTAM_TS_POSE_COARSE], st);
            rc = pose_launch_chain(ctx, ncc, &d.ctl->n_meas_coarse, d.meas, d.entry, d.pose, &g, nullptr, io);
            if (rc) return rc;
        }
    }
    // ---- fine stage :571-643 ----
    if (prof) hipEventRecord(t->ev[PTAM_TS_SEARCH_FINE], st);
    hipLaunchKernelGGL(tm_search_kernel, dim3(std::max(1, (n + 3) / 4)), dim3(256), 0, st, ctx->cam, cur->L, d, 1, 0u, 0);
    if (prof) hipEventRecord(t->ev[PTAM_TS_GATHER_FINE], st);
    const unsigned long long seq = ++t->seq;
    t->last_stream = st;
    {
        ptam_gn_opts g;
        ptam_gn_opts_default(&g);       // fine schedule :613-643
        g.estimator = o.estimator;
        PoseChainIo io{};
        io.depth_out = d.ctl->depth;
        io.result_pose = t->mbox_dev->res.pose;      // the loop's last act: pose + depth sums + sequence word into the mailbox
        io.result_depth = t->mbox_dev->depth3;
        io.result_seq = &t->mbox_dev->seq;
        io.seq = seq;
        bool fused = !no_fuse;
        if (fused) {
            // bookkeeping + the ten fine iterations in one launch.  The iteration set holds at most max(MaxPatchesPerFrame, coarse
            // + top-level set) slots: with more than the kernel's 1024 it says so instead (POSE_CHAIN_LONG) and the gather pass
            // and the general kernel follow
            if (prof) hipEventRecord(t->ev[PTAM_TS_POSE_FINE], st);
            int thr;
            const tm_pose_fn fn = tm_pose_pick(std::min(std::max(n, 1), GS_LIMIT), &thr);
            hipLaunchKernelGGL(fn, dim3(1), dim3(thr), 0, st, ctx->cam, d, 1, o.coarse_min, t->mbox_dev, g, io, d_upd, n > GS_LIMIT ? seq : 0ull);
        } else {
            hipLaunchKernelGGL(tm_gather_kernel, dim3(std::max(1, (n + TM_GATHER_THREADS - 1) / TM_GATHER_THREADS)), dim3(TM_GATHER_THREADS), 0, st, d, 1, o.coarse_subpix_its, o.coarse_min, t->mbox_dev);
            if (prof) hipEventRecord(t->ev[PTAM_TS_POSE_FINE], st);
            rc = pose_launch_chain(ctx, std::max(n, 1), &d.ctl->n_meas, d.meas, d.entry, d.pose, &g, d.outlier, io, 1);
            if (rc) return rc;
        }
        if (prof) hipEventRecord(t->ev[PTAM_TS_COUNT], st);
        HIP_TRY(hipGetLastError());
        // the frame's last kernel publishes the sequence number — or, for a list of more than 1024 entries, asks for the
        // general path, which then publishes it
        for (int pass = 0; pass < 3; pass++) {
            unsigned spins = 0;
            unsigned long long v;
            while (((v = *(volatile unsigned long long*)&t->mbox->seq) & ~POSE_CHAIN_LONG) != seq) {
                if (++spins == 100000) {
                    spins = 0;
                    const hipError_t q = hipStreamQuery(st);
                    if (q != hipSuccess && q != hipErrorNotReady) {
                        ptam_set_error("track_map: stream failed: %s", hipGetErrorString(q));
                        return PTAM_E_HIP;
                    }
                    if (q == hipSuccess && (*(volatile unsigned long long*)&t->mbox->seq & ~POSE_CHAIN_LONG) != seq) return PTAM_E_HIP;
                }
            }
            if (!(v & POSE_CHAIN_LONG)) break;
            if (pass == 2) return PTAM_E_STATE;
            t->mbox->seq = 0;   // (the stream is idle: nobody else writes the word until the next kernel does)
            if (fused) {
                // more slots than the fused kernel holds: compact them, then the register-resident kernel if the FOUND ones fit
                // (it says so otherwise: next pass)
                fused = false;
                hipLaunchKernelGGL(tm_gather_kernel, dim3(std::max(1, (n + TM_GATHER_THREADS - 1) / TM_GATHER_THREADS)), dim3(TM_GATHER_THREADS), 0, st, d, 1, o.coarse_subpix_its, o.coarse_min, t->mbox_dev);
                rc = pose_launch_chain(ctx, std::max(n, 1), &d.ctl->n_meas, d.meas, d.entry, d.pose, &g, d.outlier, io, 1);
            } else {
                rc = pose_launch_chain(ctx, std::max(n, 1), &d.ctl->n_meas, d.meas, d.entry, d.pose, &g, d.outlier, io, 2);
            }
            if (rc) return rc;
            HIP_TRY(hipGetLastError());
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    std::memcpy(out, (const void*)&t->mbox->res, sizeof *out);
    out->depth_sum = t->mbox->depth3[0];
    out->depth_sum_sq = t->mbox->depth3[1];
    out->depth_n = (int)t->mbox->depth3[2];
    if (prof) {
        HIP_TRY(hipEventSynchronize(t->ev[PTAM_TS_COUNT]));
        for (int i = 0; i < PTAM_TS_COUNT; i++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, t->ev[i], t->ev[i + 1]));
            t->stage_ms[i] += ms;
        }
        t->stage_frames++;
    }
    return PTAM_OK;
}

// Stage timing of ptam_track_map_frame (ptam_hip_bench.h): with profiling on, an event is recorded after every launch of the
// frame; the nine differences are added up per stage.  The events lengthen the frame (about 1 us per record): profiled frames
// are for the breakdown, not for the frame rate.
int ptam_tracker_set_profiling(ptam_tracker* t, int on) {
    ARG_TRY(t);
    HIP_TRY(hipSetDevice(t->ctx->device));
    if (on && !t->ev[0])
        for (int i = 0; i <= PTAM_TS_COUNT; i++) HIP_TRY(hipEventCreate(&t->ev[i]));
    t->prof = on != 0;
    for (int i = 0; i < PTAM_TS_COUNT; i++) t->stage_ms[i] = 0;
    t->stage_frames = 0;
    return PTAM_OK;
}
int ptam_tracker_stage_time(const ptam_tracker* t, int stage, double* total_ms, int* frames) {
    ARG_TRY(t && stage >= 0 && stage < PTAM_TS_COUNT && total_ms && frames);
    *total_ms = t->stage_ms[stage];
    *frames = t->stage_frames;
    return PTAM_OK;
}

int ptam_track_map(ptam_tracker* t, const ptam_kf* cur, const double pose_in[12], const ptam_trackmap_opts* opts,
                   ptam_trackmap_result* out) {
    return track_map_impl(t, const_cast<ptam_kf*>(cur), nullptr, pose_in, opts, out);
}

int ptam_track_map_frame(ptam_tracker* t, ptam_kf* cur, const uint8_t* d_frame, const double pose_in[12],
                         const ptam_trackmap_opts* opts, ptam_trackmap_result* out) {
    ARG_TRY(d_frame);
    return track_map_impl(t, cur, d_frame, pose_in, opts, out);
}

// nb frames of nb independent trackers — each with its own map, keyframe and motion-model prediction — as ONE chain of
// launches on the first tracker's queue (see TmBatchItem).  Per frame the result is what ptam_track_map_frame gives (the same
// kernel bodies on the same data).  All trackers must live on one device and share camera model, image geometry and
// halfSample variant; opts apply to every frame.
int ptam_track_map_frames_batch(int nb, ptam_tracker* const* ts, ptam_kf* const* curs, const uint8_t* const* d_frames, const double* poses_in,
                                const ptam_trackmap_opts* opts, ptam_trackmap_result* outs) {
    ARG_TRY(nb >= 1 && nb <= 4096 && ts && curs && d_frames && poses_in && outs);
    for (int i = 0; i < nb; i++) ARG_TRY(ts[i] && curs[i] && d_frames[i]);
    ptam_tracker* lead = ts[0];
    ptam_ctx* ctx = lead->ctx;
    ptam_trackmap_opts o;
    if (opts)
        o = *opts;
    else
        ptam_trackmap_opts_default(&o);
    ARG_TRY(o.max_patches >= 0 && o.coarse_subpix_its >= 0 && o.coarse_subpix_its <= 64);
    // (coarse_max / coarse_min become ints on the device and size the coarse search grid: a value past INT_MAX / 2 turned the
    //  set sizes negative and the set choice wrote outside its lists; the range is a pixel radius inside a 640-pixel image)
    ARG_TRY(o.coarse_max <= (unsigned)INT_MAX / 2 && o.coarse_min <= (unsigned)INT_MAX / 2 && o.coarse_range <= 4096u);
    ARG_TRY(o.estimator == PTAM_EST_TUKEY || o.estimator == PTAM_EST_CAUCHY || o.estimator == PTAM_EST_HUBER);
    const KfLevels& L0 = curs[0]->L;
    for (int i = 0; i < nb; i++) {
        ARG_TRY(ts[i]->ctx->device == ctx->device && curs[i]->device == ctx->device);
        ARG_TRY(ts[i]->ctx->halfsample == ctx->halfsample && std::memcmp(&ts[i]->ctx->cam, &ctx->cam, sizeof(DevCam)) == 0);
        ARG_TRY(curs[i]->n_blocks == curs[0]->n_blocks);
        for (int l = 0; l < PTAM_LEVELS; l++) ARG_TRY(curs[i]->L.w[l] == L0.w[l] && curs[i]->L.h[l] == L0.h[l]);
        for (int j = 0; j < i; j++) ARG_TRY(ts[j] != ts[i] && curs[j] != curs[i] && ts[j]->ctx != ts[i]->ctx);   // (a context's scratch serves ONE frame)
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // the other trackers' own queues: whatever they still hold (a map upload, the tail of their last frame) must be done
    for (int i = 1; i < nb; i++)
        if (ts[i]->ctx != ctx) HIP_TRY(ptam_stream_wait(ts[i]->ctx->stream));
    // ---- the argument arrays: [nb TmBatchItem | nb PoseBatchItem (coarse) | nb PoseBatchItem (fine)] ----
    const size_t b_items = (size_t)nb * sizeof(TmBatchItem), b_pose = (size_t)nb * sizeof(PoseBatchItem), b_all = b_items + 2 * b_pose;
    void* pin;
    int rc = ctx_pinned(ctx, b_all + 64, &pin);
    if (rc) return rc;
    if (lead->batch_cap < b_all) {
        HIP_TRY(ptam_stream_wait(st));
        if (lead->batch_dev) HIP_TRY(hipFree(lead->batch_dev));
        lead->batch_dev = nullptr;
        lead->batch_cap = 0;
        HIP_TRY(hipMalloc(&lead->batch_dev, b_all * 2));
        lead->batch_cap = b_all * 2;
    }
    TmBatchItem* h_it = (TmBatchItem*)pin;
    PoseBatchItem* h_pc = (PoseBatchItem*)((char*)pin + b_items);
    PoseBatchItem* h_pf = h_pc + nb;
    const TmBatchItem* d_it = (const TmBatchItem*)lead->batch_dev;
    const PoseBatchItem* d_pc = (const PoseBatchItem*)((const char*)lead->batch_dev + b_items);
    const PoseBatchItem* d_pf = d_pc + nb;
    int gx = 0, gy = 0, n_max = 0, ncc_max = 1;
    std::vector<unsigned long long> seqs((size_t)nb);
    for (int i = 0; i < nb; i++) {
        ptam_tracker* t = ts[i];
        TmBatchItem& it = h_it[i];
        std::memset(&it, 0, sizeof it);
        kf_lite_begin(curs[i], d_frames[i], &it.pa, &gx, &gy);
        it.n_pyr = gx * gy;
        it.n = t->d.n;
        it.L = curs[i]->L;
        it.d = t->d;
        std::memcpy(it.pv.v, poses_in + 12 * (size_t)i, 96);
        it.pv.use = 1;
        it.mbox = t->mbox_dev;
        const int n = t->d.n, ncc = std::max(1, std::min(n, (int)o.coarse_max));
        n_max = std::max(n_max, n);
        ncc_max = std::max(ncc_max, ncc);
        void* sst;
        double* su;
        rc = pose_chain_scratch(t->ctx, std::max(n, 1), &sst, &su);   // (sized for the fine loop; the coarse one uses its start)
        if (rc) return rc;
        PoseBatchItem& pc = h_pc[i];
        std::memset(&pc, 0, sizeof pc);
        pc.n_cap = ncc;
        pc.n_dev = &t->d.ctl->n_meas_coarse;
        pc.meas = t->d.meas;
        pc.entry = t->d.entry;
        pc.pose_io = t->d.pose;
        pc.flags = nullptr;
        pc.updates = su;
        pc.st = sst;
        pc.io.td_base = &t->d.pvs[0].proj;
        pc.io.td_index = t->d.midx;
        pc.io.td_stride = (int)sizeof(ptam_pvs_result);
        PoseBatchItem& pf = h_pf[i];
        std::memset(&pf, 0, sizeof pf);
        pf.n_cap = std::max(n, 1);
        pf.n_dev = &t->d.ctl->n_meas;
        pf.meas = t->d.meas;
        pf.entry = t->d.entry;
        pf.pose_io = t->d.pose;
        pf.flags = t->d.outlier;
        pf.updates = su;
        pf.st = sst;
        seqs[(size_t)i] = ++t->seq;
        t->last_stream = st;
        pf.io.depth_out = t->d.ctl->depth;
        pf.io.result_pose = t->mbox_dev->res.pose;
        pf.io.result_depth = t->mbox_dev->depth3;
        pf.io.result_seq = &t->mbox_dev->seq;
        pf.io.seq = seqs[(size_t)i];
    }
    HIP_TRY(hipMemcpyAsync(lead->batch_dev, pin, b_all, hipMemcpyHostToDevice, st));
    const int n_pyr = gx * gy, n_pvs = std::max(1, (n_max + 255) / 256);
    if (ctx->halfsample == PTAM_HALFSAMPLE_T)
        hipLaunchKernelGGL(tm_pyr_pvs_batch_kernel<PTAM_HALFSAMPLE_T>, dim3(n_pyr + n_pvs, nb), dim3(256), 0, st, d_it, gx, ctx->cam);
    else
        hipLaunchKernelGGL(tm_pyr_pvs_batch_kernel<PTAM_HALFSAMPLE_R>, dim3(n_pyr + n_pvs, nb), dim3(256), 0, st, d_it, gx, ctx->cam);
    kf_launch_detect_batch(nb, L0, d_it, sizeof(TmBatchItem), offsetof(TmBatchItem, L), st);
    hipLaunchKernelGGL(tm_compact_select_batch_kernel, dim3(1 + fast_compact_blocks(L0), nb), dim3(1024), 0, st, d_it, o);
    // ---- coarse stage :519-569 ----
    hipLaunchKernelGGL(tm_search_batch_kernel, dim3((ncc_max + 3) / 4, nb), dim3(256), 0, st, ctx->cam, d_it, 0, o.coarse_range, o.coarse_subpix_its);
    // (fused bookkeeping + pose loop per stage when no frame's list can outgrow the register-resident kernel: maps of at most
    //  1024 points; larger maps keep the gather pass and the small / general kernel pair)
    static const bool no_fuse = ptam_ab_env("PTAM_TM_NO_FUSE") != nullptr;
    const bool fuse = !no_fuse && n_max <= GS_LIMIT && ncc_max <= GS_LIMIT;
    {
        ptam_gn_opts g;
        ptam_gn_opts_default(&g);
        g.nonlinear_mask = 0x3ff;       // every coarse iteration re-projects (:556-562)
        g.override_sigma_sq = 1.0;      // :565
        g.mark_outliers_iter = -1;
        g.estimator = o.estimator;
        if (fuse)
            tm_pose_launch_batch(ctx, nb, ncc_max, d_it, d_pc, 0, o.coarse_min, g);
        else {
            hipLaunchKernelGGL(tm_gather_batch_kernel, dim3(std::max(1, (ncc_max + TM_GATHER_THREADS - 1) / TM_GATHER_THREADS), nb), dim3(TM_GATHER_THREADS), 0, st,
                               d_it, 0, (int)o.coarse_subpix_its, o.coarse_min);
            rc = pose_launch_chain_batch(ctx, nb, ncc_max, d_pc, &g);
            if (rc) return rc;
        }
    }
    // ---- fine stage :571-643 ----
    hipLaunchKernelGGL(tm_search_batch_kernel, dim3(std::max(1, (n_max + 3) / 4), nb), dim3(256), 0, st, ctx->cam, d_it, 1, 0u, 0);
    {
        ptam_gn_opts g;
        ptam_gn_opts_default(&g);       // fine schedule :613-643
        g.estimator = o.estimator;
        if (fuse)
            tm_pose_launch_batch(ctx, nb, std::max(n_max, 1), d_it, d_pf, 1, o.coarse_min, g);
        else {
            hipLaunchKernelGGL(tm_gather_batch_kernel, dim3(std::max(1, (n_max + TM_GATHER_THREADS - 1) / TM_GATHER_THREADS), nb), dim3(TM_GATHER_THREADS), 0, st,
                               d_it, 1, (int)o.coarse_subpix_its, o.coarse_min);
            rc = pose_launch_chain_batch(ctx, nb, std::max(n_max, 1), d_pf, &g);
            if (rc) return rc;
        }
    }
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < nb; i++) {
        ptam_tracker* t = ts[i];
        const unsigned long long seq = seqs[(size_t)i];
        unsigned spins = 0;
        while (*(volatile unsigned long long*)&t->mbox->seq != seq) {
            if (++spins == 100000) {
                spins = 0;
                const hipError_t q = hipStreamQuery(st);
                if (q != hipSuccess && q != hipErrorNotReady) {
                    ptam_set_error("track_map_frames_batch: stream failed: %s", hipGetErrorString(q));
                    return PTAM_E_HIP;
                }
                if (q == hipSuccess && *(volatile unsigned long long*)&t->mbox->seq != seq) return PTAM_E_HIP;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        ptam_trackmap_result* out = outs + i;
        std::memcpy(out, (const void*)&t->mbox->res, sizeof *out);
        out->depth_sum = t->mbox->depth3[0];
        out->depth_sum_sq = t->mbox->depth3[1];
        out->depth_n = (int)t->mbox->depth3[2];
    }
    return PTAM_OK;
}

int ptam_tracker_read_iteration_set(ptam_tracker* t, ptam_trackmap_meas* out, int cap, int* n_out) {
    ARG_TRY(t && n_out);
    ptam_ctx* ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    TmCtl c;
    HIP_TRY(hipMemcpy(&c, t->d.ctl, sizeof c, hipMemcpyDeviceToHost));
    const int ns = c.n_slots;
    *n_out = ns;
    if (!out || ns == 0) return PTAM_OK;
    std::vector<int> list((size_t)ns), sf((size_t)ns), ss((size_t)ns), ms((size_t)std::max(c.n_meas, 1)), ou((size_t)std::max(c.n_meas, 1));
    std::vector<ptam_patch_query> q((size_t)ns);
    std::vector<ptam_template_result> tr((size_t)ns);
    std::vector<double2> v2((size_t)ns);
    HIP_TRY(hipMemcpy(list.data(), t->d.list, (size_t)ns * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sf.data(), t->d.slot_found, (size_t)ns * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ss.data(), t->d.slot_subpix, (size_t)ns * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(q.data(), t->d.q, (size_t)ns * sizeof(ptam_patch_query), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(v2.data(), t->d.slot_v2, (size_t)ns * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tr.data(), t->d.tres, (size_t)ns * sizeof(ptam_template_result), hipMemcpyDeviceToHost));
    if (c.n_meas > 0) {
        HIP_TRY(hipMemcpy(ms.data(), t->d.mslot, (size_t)c.n_meas * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ou.data(), t->d.outlier, (size_t)c.n_meas * 4, hipMemcpyDeviceToHost));
    }
    for (int s = 0; s < ns && s < cap; s++) {
        ptam_trackmap_meas& m = out[s];
        m.point = list[(size_t)s];
        m.level = tr[(size_t)s].bad ? -1 : q[(size_t)s].level;   // (a bad template: the point was dropped before the search)
        m.found = sf[(size_t)s];
        m.did_subpix = ss[(size_t)s];
        m.outlier = 0;
        m.pad_ = 0;
        m.v2_found[0] = v2[(size_t)s].x;
        m.v2_found[1] = v2[(size_t)s].y;
    }
    if (c.outlier_by_slot) {   // (fused pose kernel: the flags sit at the slots)
        std::vector<int> os((size_t)ns);
        HIP_TRY(hipMemcpy(os.data(), t->d.outlier, (size_t)ns * 4, hipMemcpyDeviceToHost));
        for (int s = 0; s < ns && s < cap; s++) out[s].outlier = sf[(size_t)s] ? os[(size_t)s] : 0;
        return PTAM_OK;
    }
    for (int k = 0; k < c.n_meas; k++)
        if (ms[(size_t)k] < cap) out[ms[(size_t)k]].outlier = ou[(size_t)k];
    return PTAM_OK;
}

int ptam_refind_batch(ptam_ctx* ctx, const ptam_kf* kf, const double kf_pose[12], int n, const ptam_pvs_point* points,
                      const ptam_template_query* sources, ptam_refind_result* out) {
    ARG_TRY(ctx && kf && kf_pose && n >= 0);
    if (n == 0) return PTAM_OK;
    ARG_TRY(points && sources && out);
    ARG_TRY(kf->device == ctx->device);
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<TmSrc> hs((size_t)n);
    for (int i = 0; i < n; i++) {
        const ptam_template_query& q = sources[i];
        ARG_TRY(q.src_kf && q.src_level >= 0 && q.src_level < PTAM_LEVELS && q.src_kf->device == ctx->device);
        hs[(size_t)i].im = q.src_kf->L.im[q.src_level];
        hs[(size_t)i].w = q.src_kf->L.w[q.src_level];
        hs[(size_t)i].h = q.src_kf->L.h[q.src_level];
        hs[(size_t)i].cx = q.center_x;
        hs[(size_t)i].cy = q.center_y;
    }
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t N = (size_t)n;
    const size_t o_pts = take(N * sizeof(ptam_pvs_point)), o_src = take(N * sizeof(TmSrc)), o_pose = take(96),
                 o_jobs = take(N * sizeof(TemplateJob)), o_tm = take(N * 64), o_tr = take(N * sizeof(ptam_template_result)),
                 o_q = take(N * sizeof(ptam_patch_query)), o_r = take(N * sizeof(ptam_patch_result)),
                 o_sr = take(N * sizeof(ptam_subpix_result)), o_out = take(N * sizeof(ptam_refind_result));
    void* s;
    int rc = ctx_scratch(ctx, off, &s);
    if (rc) return rc;
    char* b = (char*)s;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(b + o_pts, points, N * sizeof(ptam_pvs_point), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + o_src, hs.data(), N * sizeof(TmSrc), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + o_pose, kf_pose, 96, hipMemcpyHostToDevice, st));
    const int g = (n + 255) / 256;
    TemplateJob* d_jobs = (TemplateJob*)(b + o_jobs);
    ptam_patch_query* d_q = (ptam_patch_query*)(b + o_q);
    ptam_patch_result* d_r = (ptam_patch_result*)(b + o_r);
    ptam_subpix_result* d_sr = (ptam_subpix_result*)(b + o_sr);
    ptam_template_result* d_tr = (ptam_template_result*)(b + o_tr);
    uint8_t* d_tm = (uint8_t*)(b + o_tm);
    hipLaunchKernelGGL(refind_prep_kernel, dim3(g), dim3(256), 0, st, ctx->cam, n, (const ptam_pvs_point*)(b + o_pts), (const TmSrc*)(b + o_src),
                       (const double*)(b + o_pose), d_jobs, d_q);
    rc = patch_launch_templates_dev(ctx, n, d_jobs, d_tm, d_tr, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(refind_mask_kernel, dim3(g), dim3(256), 0, st, n, (const ptam_template_result*)d_tr, d_q);
    rc = patch_launch_search_dev(ctx, kf, n, d_q, d_tm, d_r, nullptr, nullptr);
    if (rc) return rc;
    rc = patch_launch_subpix_dev(ctx, kf, n, d_q, d_r, d_tm, d_sr, nullptr, 8);
    if (rc) return rc;
    hipLaunchKernelGGL(refind_finish_kernel, dim3(g), dim3(256), 0, st, n, (const ptam_patch_query*)d_q, (const ptam_patch_result*)d_r,
                       (const ptam_subpix_result*)d_sr, (ptam_refind_result*)(b + o_out));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, b + o_out, N * sizeof(ptam_refind_result), hipMemcpyDeviceToHost, st));
    HIP_TRY(ptam_stream_wait(st));   // (also keeps hs[] alive until its pageable copy has been staged)
    return PTAM_OK;
}

struct ptam_refinder {
    ptam_ctx* ctx;
    RpFinder* st;   // device
};
int ptam_refinder_create(ptam_ctx* ctx, ptam_refinder** out) {
    ARG_TRY(ctx && out);
    HIP_TRY(hipSetDevice(ctx->device));
    ptam_refinder* f = new ptam_refinder();
    f->ctx = ctx;
    if (hipMalloc((void**)&f->st, sizeof(RpFinder)) != hipSuccess || hipMemset(f->st, 0, sizeof(RpFinder)) != hipSuccess) {
        if (f->st) hipFree(f->st);
        delete f;
        ptam_set_error("ptam_refinder_create: allocation failed");
        return PTAM_E_HIP;
    }
    *out = f;
    return PTAM_OK;
}
int ptam_refinder_destroy(ptam_refinder* f) {
    if (!f) return PTAM_OK;
    hipSetDevice(f->ctx->device);
    ptam_stream_wait(f->ctx->stream);
    hipFree(f->st);
    delete f;
    return PTAM_OK;
}
int ptam_refind_pairs(ptam_ctx* ctx, ptam_refinder* finder, int n, const ptam_refind_pair* pairs, ptam_refind_result* out,
                      int32_t* template_kept) {
    ARG_TRY(ctx && finder && finder->ctx->device == ctx->device && n >= 0);
    if (n == 0) return PTAM_OK;
    ARG_TRY(pairs && out);
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<RpPair> hp((size_t)n);
    std::vector<KfLevels> hl;
    std::vector<const ptam_kf*> seen;
    for (int i = 0; i < n; i++) {
        const ptam_refind_pair& p = pairs[i];
        const ptam_template_query& q = p.source;
        ARG_TRY(p.kf && p.kf->device == ctx->device);
        ARG_TRY(q.src_kf && q.src_level >= 0 && q.src_level < PTAM_LEVELS && q.src_kf->device == ctx->device);
        RpPair& r = hp[(size_t)i];
        r.pt = p.point;
        r.src.im = q.src_kf->L.im[q.src_level];
        r.src.w = q.src_kf->L.w[q.src_level];
        r.src.h = q.src_kf->L.h[q.src_level];
        r.src.cx = q.center_x;
        r.src.cy = q.center_y;
        std::memcpy(r.pose, p.kf_pose, 96);
        r.id = (long long)p.point_id;
        r.skip = p.skip != 0;
        int k = -1;   // (lists name few keyframes, mostly in runs: a linear search from the back finds the last one at once)
        for (int s = (int)seen.size() - 1; s >= 0; s--)
            if (seen[(size_t)s] == p.kf) {
                k = s;
                break;
            }
        if (k < 0) {
            k = (int)seen.size();
            seen.push_back(p.kf);
            hl.push_back(p.kf->L);
        }
        r.kf = k;
    }
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t N = (size_t)n;
    const size_t o_pairs = take(N * sizeof(RpPair)), o_ls = take(hl.size() * sizeof(KfLevels)), o_jobs = take(N * sizeof(TemplateJob)),
                 o_q = take(N * sizeof(ptam_patch_query)), o_m2 = take(N * 32), o_reach = take(N * 4), o_det = take(N * 4), o_R = take(N * 4),
                 o_nr = take(4), o_ref = take(N * 4), o_src = take(N * 4), o_dacc = take(N * 4), o_bad = take(N * 4), o_tm = take(N * 64),
                 o_tr = take(N * sizeof(ptam_template_result)), o_out = take(N * sizeof(ptam_refind_result)), o_kept = take(N * 4);
    void* sc;
    int rc = ctx_scratch(ctx, off, &sc);
    if (rc) return rc;
    char* b = (char*)sc;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(b + o_pairs, hp.data(), N * sizeof(RpPair), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b + o_ls, hl.data(), hl.size() * sizeof(KfLevels), hipMemcpyHostToDevice, st));
    RpDev d;
    d.n = n;
    d.pairs = (const RpPair*)(b + o_pairs);
    d.Ls = (const KfLevels*)(b + o_ls);
    d.jobs = (TemplateJob*)(b + o_jobs);
    d.q = (ptam_patch_query*)(b + o_q);
    d.m2 = (double*)(b + o_m2);
    d.reach = (int*)(b + o_reach);
    d.detbad = (int*)(b + o_det);
    d.R = (int*)(b + o_R);
    d.nr = (int*)(b + o_nr);
    d.refresh = (int*)(b + o_ref);
    d.srcp = (int*)(b + o_src);
    d.dacc = (int*)(b + o_dacc);
    d.bad = (int*)(b + o_bad);
    d.tm = (uint8_t*)(b + o_tm);
    d.tres = (ptam_template_result*)(b + o_tr);
    d.st = finder->st;
    d.out = (ptam_refind_result*)(b + o_out);
    d.kept = (int*)(b + o_kept);
    hipLaunchKernelGGL(rp_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ctx->cam, d);
    hipLaunchKernelGGL(rp_scan_kernel, dim3(1), dim3(1024), 0, st, d);
    hipLaunchKernelGGL(rp_template_kernel, dim3((n + 3) / 4), dim3(256), 0, st, d);
    hipLaunchKernelGGL(rp_search_kernel, dim3((n + 3) / 4), dim3(256), 0, st, d);
    hipLaunchKernelGGL(rp_state_kernel, dim3(1), dim3(64), 0, st, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d.out, N * sizeof(ptam_refind_result), hipMemcpyDeviceToHost, st));
    if (template_kept) HIP_TRY(hipMemcpyAsync(template_kept, d.kept, N * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ptam_stream_wait(st));   // (also keeps the staging vectors alive until their pageable copies have been staged)
    return PTAM_OK;
}

// Measurement helper (like ptam_ba_bench_jacobian): the "replicas" axis of the tracking path driven natively.
// n independent trackers — each with its own context (stream, scratch, mailbox), map and keyframes — are run by n host
// threads, frames_each frames per thread: per frame the two permutations are handed over (ptam_tracker_set_shuffle) and
// ptam_track_map_frame is called exactly as the reference's tracker thread would.  The threads start together; *seconds_out
// is the wall time from that start to the last thread's return, n * frames_each frames in all.
int ptam_bench_track_frames(int n, ptam_tracker* const* trackers, ptam_kf* const* current, const uint8_t* const* d_frames,
                            const double pose_in[12], const ptam_trackmap_opts* opts, const int32_t* shuffle_levels,
                            const int32_t* shuffle_fine, int frames_each, double* seconds_out) {
    ARG_TRY(n >= 1 && n <= 1024 && trackers && current && d_frames && pose_in && shuffle_levels && shuffle_fine && frames_each >= 1 && seconds_out);
    for (int i = 0; i < n; i++) ARG_TRY(trackers[i] && current[i] && d_frames[i]);
    std::atomic<int> ready{0}, failed{0};
    std::atomic<bool> go{false};
    std::vector<std::string> errs((size_t)n);
    std::vector<std::thread> th;
    th.reserve((size_t)n);
    for (int i = 0; i < n; i++)
        th.emplace_back([&, i]() {
            ptam_trackmap_result res;
            ready.fetch_add(1);
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            for (int f = 0; f < frames_each; f++) {
                int rc = ptam_tracker_set_shuffle(trackers[i], shuffle_levels, shuffle_fine);
                if (!rc) rc = ptam_track_map_frame(trackers[i], current[i], d_frames[i], pose_in, opts, &res);
                if (rc) {
                    errs[(size_t)i] = ptam_last_error();
                    failed.store(rc);
                    return;
                }
            }
        });
    while (ready.load() < n) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto& t : th) t.join();
    *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failed.load()) {
        for (const auto& e : errs)
            if (!e.empty()) {
                ptam_set_error("ptam_bench_track_frames: a worker failed: %s", e.c_str());
                break;
            }
        return failed.load();
    }
    return PTAM_OK;
}

// Measurement helper: `rounds` rounds of nb frames as batches (ptam_track_map_frames_batch).  groups == 1: one host thread, one
// batch of nb per round.  groups > 1: the trackers are dealt into that many groups, each driven by its own host thread on its
// own queue (the group's first tracker leads), so that one group's single-workgroup pose loops run beside another group's
// searches.  Per round every tracker is handed its permutations first, as a caller tracking nb cameras would.
// *seconds_out = wall time from the common start to the last thread's return.
int ptam_bench_track_batch(int nb, ptam_tracker* const* trackers, ptam_kf* const* current, const uint8_t* const* d_frames,
                           const double pose_in[12], const ptam_trackmap_opts* opts, const int32_t* shuffle_levels,
                           const int32_t* shuffle_fine, int rounds, int groups, double* seconds_out) {
    ARG_TRY(nb >= 1 && nb <= 4096 && trackers && current && d_frames && pose_in && shuffle_levels && shuffle_fine && rounds >= 1 && seconds_out);
    ARG_TRY(groups >= 1 && groups <= nb);
    std::atomic<int> ready{0}, failed{0};
    std::atomic<bool> go{false};
    std::vector<std::string> errs((size_t)groups);
    auto work = [&](int g) {
        const int per = (nb + groups - 1) / groups, i0 = g * per, i1 = std::min(nb, i0 + per), m = i1 - i0;
        if (m <= 0) return;
        std::vector<double> poses((size_t)m * 12);
        for (int i = 0; i < m; i++) std::memcpy(&poses[(size_t)i * 12], pose_in, 96);
        std::vector<ptam_trackmap_result> res((size_t)m);
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        for (int r = 0; r < rounds; r++) {
            int rc = PTAM_OK;
            for (int i = i0; i < i1 && !rc; i++) rc = ptam_tracker_set_shuffle(trackers[i], shuffle_levels, shuffle_fine);
            if (!rc) rc = ptam_track_map_frames_batch(m, trackers + i0, current + i0, d_frames + i0, poses.data(), opts, res.data());
            if (rc) {
                errs[(size_t)g] = ptam_last_error();
                failed.store(rc);
                return;
            }
        }
    };
    std::vector<std::thread> th;
    int n_threads = 0;
    for (int g = 0; g < groups; g++)
        if (g * ((nb + groups - 1) / groups) < nb) {
            th.emplace_back(work, g);
            n_threads++;
        }
    while (ready.load() < n_threads && !failed.load()) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto& t : th) t.join();
    *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failed.load()) {
        for (const auto& e : errs)
            if (!e.empty()) {
                ptam_set_error("ptam_bench_track_batch: a group failed: %s", e.c_str());
                break;
            }
        return failed.load();
    }
    return PTAM_OK;
}

}   // extern "C"

int pose_hazards_read_trackmap(hipStream_t st, unsigned long long* out) {   // (this translation unit's copy of the counter)
    HIP_TRY(hipMemcpyFromSymbolAsync(out, HIP_SYMBOL(g_pose_hazards), 8, 0, hipMemcpyDeviceToHost, st));
    return PTAM_OK;
}
void trackmap_preload_kernels() {
    ptam_preload((const void*)refind_prep_kernel);
    ptam_preload((const void*)refind_mask_kernel);
    ptam_preload((const void*)refind_finish_kernel);
    ptam_preload((const void*)rp_prep_kernel);
    ptam_preload((const void*)rp_scan_kernel);
    ptam_preload((const void*)rp_template_kernel);
    ptam_preload((const void*)rp_search_kernel);
    ptam_preload((const void*)rp_state_kernel);
    ptam_preload((const void*)tm_select_kernel);
    ptam_preload((const void*)tm_pyr_pvs_kernel<PTAM_HALFSAMPLE_R>);
    ptam_preload((const void*)tm_pyr_pvs_kernel<PTAM_HALFSAMPLE_T>);
    ptam_preload((const void*)tm_compact_select_kernel);
    ptam_preload((const void*)tm_kf_pvs_kernel<PTAM_HALFSAMPLE_R>);
    ptam_preload((const void*)tm_kf_pvs_kernel<PTAM_HALFSAMPLE_T>);
    ptam_preload((const void*)tm_search_kernel);
    ptam_preload((const void*)tm_gather_kernel);
    ptam_preload((const void*)tm_finder_carry_kernel);
    ptam_preload((const void*)tm_pose_kernel<1, GS_WAVE_LIMIT>);
    ptam_preload((const void*)tm_pose_kernel<1, GS_THREADS>);
    ptam_preload((const void*)tm_pose_kernel<GS_MPT, GS_THREADS>);
    ptam_preload((const void*)tm_pose_batch_kernel<1, GS_WAVE_LIMIT>);
    ptam_preload((const void*)tm_pose_batch_kernel<1, GS_THREADS>);
    ptam_preload((const void*)tm_pose_batch_kernel<GS_MPT, GS_THREADS>);
    ptam_preload((const void*)tm_pyr_pvs_batch_kernel<PTAM_HALFSAMPLE_R>);
    ptam_preload((const void*)tm_pyr_pvs_batch_kernel<PTAM_HALFSAMPLE_T>);
    ptam_preload((const void*)tm_compact_select_batch_kernel);
    ptam_preload((const void*)tm_search_batch_kernel);
    ptam_preload((const void*)tm_gather_batch_kernel);
}
