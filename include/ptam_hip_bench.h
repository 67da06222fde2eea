/* ptam_hip_bench.h — measurement-only entry points of libptam_hip.so.
 *
 * Nothing here is part of the reference's surface or of the drop-in boundary (include/ptam_hip.h): these calls exist so
 * that bench.py and tools/ can time one kernel in isolation or drive frames from native host threads.  A caller that
 * replaces the reference's Tracker / MapMaker / Bundle never needs this header. */
#ifndef PTAM_HIP_BENCH_H
#define PTAM_HIP_BENCH_H
#include "ptam_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Measurement helper, not part of the reference's surface: n independent trackers (each with its own context, map and
 * keyframes) driven by n host threads inside the library, frames_each frames per thread — per frame ptam_tracker_set_shuffle
 * then ptam_track_map_frame, as the tracker thread of src/Tracker.cc:442-696 would issue them.  *seconds_out = wall time
 * from the common start to the last return (bench.py: aggregate frames/s of replicas on one device). */
int ptam_bench_track_frames(int n, ptam_tracker* const* trackers, ptam_kf* const* current, const uint8_t* const* d_frames,
                            const double pose_in[12], const ptam_trackmap_opts* opts, const int32_t* shuffle_levels,
                            const int32_t* shuffle_fine, int frames_each, double* seconds_out);
/* Measurement helper: `rounds` rounds of ptam_tracker_set_shuffle (every tracker) + ptam_track_map_frames_batch.  groups == 1:
 * one host thread, one batch of nb per round; groups > 1: the trackers dealt into that many groups, each batched by its own
 * host thread on its own queue.  *seconds_out = wall time (bench.py: frames/s of nb cameras tracked as batches). */
int ptam_bench_track_batch(int nb, ptam_tracker* const* trackers, ptam_kf* const* current, const uint8_t* const* d_frames,
                           const double pose_in[12], const ptam_trackmap_opts* opts, const int32_t* shuffle_levels,
                           const int32_t* shuffle_fine, int rounds, int groups, double* seconds_out);
/* One camera tracked over a SEQUENCE of frames, closed loop, from one native host thread: per frame ptam_tracker_set_shuffle
 * + ptam_track_frame (keyframe of the frame, motion-model prediction, bTryCoarse heuristics, TrackMap, motion-model update —
 * src/Tracker.cc:94,134-137).  The n_frames device-resident frames are visited in order, `passes` times over (a closed
 * trajectory: frame 0 follows frame n_frames - 1); the model starts at *m and is left at the last tracked pose.
 * stats_out (nullable, 8 doubles): frames, sum of searched patches (n_coarse + n_top + n_fine), sum of templates_reused,
 * sum of n_meas, frames with mbDidCoarse, frames whose bTryCoarse heuristic said yes, max over frames of |t_tracked - t_true|
 * (0 unless poses_true is given: n_frames x 12), frames with fewer than 50 measurements. */
int ptam_bench_track_sequence(ptam_tracker* t, ptam_kf* current, int n_frames, const uint8_t* const* d_frames,
                              ptam_motion_model* m, const ptam_trackmap_opts* opts, const int32_t* shuffle_levels,
                              const int32_t* shuffle_fine, int passes, const double* poses_true, double* seconds_out,
                              double* stats_out);

/* Stage timing of ptam_track_map_frame: with profiling on, a HIP event is recorded on the tracker's queue after every launch of
 * the frame (the records lengthen the frame: use profiled frames for the breakdown only).  stage_time returns the sum over the
 * profiled frames of the time between the stage's launch and the next one's (kernel + its boundary). */
enum {
    PTAM_TS_PYR_PVS = 0,        /* pyramid of the new frame + PVS pass over the map (one launch) */
    PTAM_TS_DETECT = 1,         /* FAST-10 on the four levels */
    PTAM_TS_COMPACT_SELECT = 2, /* corner compaction + row LUTs + choice of the search sets */
    PTAM_TS_SEARCH_COARSE = 3,  /* coarse stage: templates + range-30 search + sub-pixel */
    PTAM_TS_GATHER_COARSE = 4,
    PTAM_TS_POSE_COARSE = 5,    /* ten coarse pose iterations */
    PTAM_TS_SEARCH_FINE = 6,    /* re-projection + templates + search + sub-pixel of the top-level remainder and the fine set */
    PTAM_TS_GATHER_FINE = 7,
    PTAM_TS_POSE_FINE = 8,      /* ten fine pose iterations + result publication */
    PTAM_TS_COUNT = 9
};
int ptam_tracker_set_profiling(ptam_tracker* t, int on);
int ptam_tracker_stage_time(const ptam_tracker* t, int stage, double* total_ms, int* frames);

/* launch ONLY the K7 kernel `reps` times on the prepared problem (pass 1 + sigma must have run
 * once: done internally), HIP-event timed; returns average ms per launch and the algorithmic
 * byte count of one launch. */
int ptam_ba_bench_jacobian(ptam_ba* ba, int reps, double* avg_ms, double* algorithmic_bytes);
/* the same bracket over `n` bundles of one context (copies of one problem) launched round-robin: with
 * (n - 1) working sets larger than the 256 MB Infinity Cache every launch finds its data in HBM only. */
int ptam_ba_bench_jacobian_rotating(ptam_ba** bas, int n, int reps, double* avg_ms);
/* Test hook (host only, no device needed): the index map the Schur tile kernel's epilogue gathers a partial tile through
 * (csrc/ba_schur.inc: schur_index_map) — for variant 0..5 = {diagonal pair with 3 / 2 / 1 row fragments, off-diagonal pair with
 * 3 / 2 / 1 row fragments against 3}, out[e] = value index * 64 + lane of element e of the tile's [8][8][6][6] + E[48] layout in
 * the accumulator layout of the cross-wave reduction, 0xffff for an element nothing is multiplied into.  Returns the number of
 * elements (2352) or a negative error. */
int ptam_ba_schur_index_map(int variant, uint16_t* out, int cap);
/* Test hook: one of the index structures ptam_ba_prepare built on the device (csrc/ba_prepare.inc), copied to `out` (at most
 * cap_bytes).  Returns the structure's size in bytes (also with out == NULL: a size query) or a negative error; the bundle must be
 * prepared.  tests/test_gpu_prepare.py rebuilds every one of them with numpy (and the work split with csrc/ba_split.h compiled
 * for the host) and compares. */
enum {
    PTAM_BL_COUNTS = 0,      /* int32[16]: C, F, P, M, band, n_chunks, n_tiles, n_pairs, n_schur_wg, n_schur_entries, n_segments, grid_acc */
    PTAM_BL_ROWPTR = 1,      /* int32[P + 1] */
    PTAM_BL_M_CAM = 2,       /* int32[M] camera of the sorted measurements */
    PTAM_BL_M_PT = 3,        /* int32[M] dense point id */
    PTAM_BL_M_ORIG = 4,      /* int32[M] insertion index */
    PTAM_BL_M_FIDX = 5,      /* int32[M] free-camera index or -1 */
    PTAM_BL_M_FOUND = 6,     /* double[M][2] */
    PTAM_BL_M_S = 7,         /* double[M] dSqrtInvNoise */
    PTAM_BL_PT_ORIG = 8,     /* int32[P] dense point id -> original id */
    PTAM_BL_POINTS = 9,      /* double[P][3] current positions of the dense points */
    PTAM_BL_CHUNKS = 10,     /* int32[n_chunks][4] pt_begin, pt_end, m_begin, m_end */
    PTAM_BL_S_ENTRIES = 11,  /* int32[n_schur_entries][8] pt, ma, mb, cost | pattern << 16, offa[2], offb[2] */
    PTAM_BL_S_SEGS = 12,     /* int32[n_segments][4] pair, e_begin, e_end, slot */
    PTAM_BL_S_WG_SEG = 13,   /* int32[n_schur_wg + 1] */
    PTAM_BL_S_PAIR_BEGIN = 14, /* int32[n_pairs + 1] */
    PTAM_BL_S_WG_HEAD = 15,  /* int32[n_schur_wg][8] */
    PTAM_BL_CAM_PTR = 16,    /* deterministic mode: int32[tiles of 1024 measurements][F + 1] */
    PTAM_BL_CAM_MEAS = 17    /* deterministic mode: int32[measurements by free cameras] */
};
int ptam_ba_debug_lists(ptam_ba* ba, int which, void* out, size_t cap_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PTAM_HIP_BENCH_H */
