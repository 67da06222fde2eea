// track_internal.h — launch helpers the resident TrackMap chain (trackmap.hip) uses from the other translation units.
// Every helper only enqueues on the context's stream; slot ranges and counts live in DEVICE memory (d_range = {begin, end},
// d_n), because which patches are searched and how many are found is decided on the device (no host round trip inside a
// tracked frame).
#pragma once
#include "common.h"
#include "keyframe.h"

struct TemplateJob {          // device-side form of ptam_template_query (keyframe handle resolved to its level image)
    const uint8_t* im;
    int w, h;
    int search_level;
    int cx, cy;
    double wi[4];
};

// patch.hip
int patch_launch_templates_dev(ptam_ctx* ctx, int n_cap, const TemplateJob* d_jobs, uint8_t* d_tmpl, ptam_template_result* d_res,
                               const int* d_range);
// d_tres (nullable): a query whose template came out bad (Finder.TemplateBad(), src/Tracker.cc:876) is not searched
int patch_launch_search_dev(ptam_ctx* ctx, const ptam_kf* kf, int n_cap, const ptam_patch_query* d_q, const uint8_t* d_tmpl,
                            ptam_patch_result* d_r, const int* d_range, const ptam_template_result* d_tres);
// sub-pixel refinement of the patches the coarse search found (queries taken from the search's own queries / results)
int patch_launch_subpix_dev(ptam_ctx* ctx, const ptam_kf* kf, int n_cap, const ptam_patch_query* d_q, const ptam_patch_result* d_pr,
                            const uint8_t* d_tmpl, ptam_subpix_result* d_sr, const int* d_range, int max_its);
// keyframe.hip: KeyFrame::MakeKeyFrame_Lite of a device-resident frame, enqueued on `stream`
int kf_make_lite_on(ptam_ctx* ctx, ptam_kf* kf, const uint8_t* d_im, hipStream_t stream);
// the same in pieces, for a caller that fuses the pyramid and the compaction into launches of its own (keyframe_device.h)
struct PyrArgs;
void kf_lite_begin(ptam_kf* kf, const uint8_t* d_src, PyrArgs* a_out, int* gx, int* gy);
void kf_launch_detect(ptam_kf* kf, hipStream_t stream);
// FAST detection for nb keyframes of equal geometry in one launch: their KfLevels sit at d_items + i * stride + off_levels
void kf_launch_detect_batch(int nb, const KfLevels& L, const void* d_items, size_t stride, size_t off_levels, hipStream_t stream);   // (L: the common geometry)
// pvs.hip
struct PoseArg {   // a pose handed over by value
    double v[12];
    int use;
};
// host_pose (nullable): the pose as a kernel argument, also written to d_pose; otherwise d_pose is read
// d_finder_bad (nullable): word i * finder_stride bytes further on is set when point i's warp is rejected (PatchFinder::mbTemplateBad)
int pvs_launch_dev(ptam_ctx* ctx, int n, const ptam_pvs_point* d_pts, double* d_pose, const double* host_pose, ptam_pvs_result* d_out,
                   int* d_finder_bad = nullptr, int finder_stride = 0);

// pose.hip: the ten-iteration loop on a measurement list whose length sits in device memory, with the extras of the chain
struct PoseChainIo {
    // TrackerData state of the measurements when the loop ends (v3Cam, v2Image, m2CamDerivs as left by the last
    // ProjectAndDerivs / LinearUpdate), scattered to td_base + td_index[i] * td_stride bytes (ptam_projection each); null: no
    void* td_base;
    const int* td_index;
    int td_stride;
    // scene depth statistics over the measurements (src/Tracker.cc:680-690): {sum z, sum z^2, count}; null: no
    double* depth_out;
    // the frame's result block in host-mapped memory: the loop's last act is to publish the pose (12 doubles at
    // result_pose), the depth sums (3 doubles at result_depth) and then the sequence word the host spins on; null: no
    double* result_pose;
    double* result_depth;
    unsigned long long* result_seq;
    unsigned long long seq;
};
// one frame of a batch (ptam_track_map_frames_batch): what pose_launch_chain takes as arguments, in device memory
struct PoseBatchItem {
    int n_cap;                      // capacity of the list (its length sits at n_dev)
    const int* n_dev;
    const ptam_pose_meas* meas;
    const ptam_projection* entry;
    double* pose_io;
    int32_t* flags;                 // outlier flags (nullable)
    double* updates;                // scratch: the iterations' updates
    void* st;                       // scratch of the general kernel (lists of more than 1024 measurements)
    PoseChainIo io;
};
int pose_launch_chain_batch(ptam_ctx* ctx, int nb, int n_cap_max, const PoseBatchItem* d_items, const ptam_gn_opts* opts);
int pose_chain_scratch(ptam_ctx* ctx, int n_cap, void** st_out, double** updates_out);
#define POSE_CHAIN_LONG (1ull << 63)   // in the published sequence word: "the list is longer than the register-resident kernel holds" (pose.hip)
int pose_launch_chain(ptam_ctx* ctx, int n_cap, const int* d_n, const ptam_pose_meas* d_meas, const ptam_projection* d_entry,
                      double* d_pose_inout, const ptam_gn_opts* opts, int32_t* d_outlier_flags, const PoseChainIo& io, int mode = 0);
