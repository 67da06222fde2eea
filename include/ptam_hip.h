/*
 * ptam_hip.h — C ABI of libptam_hip.so: PTAM's tracking + bundle-adjustment hot path on
 * MI355X (gfx950), hand-written HIP kernels behind plain-C entry points.
 *
 * The reference (cggos/ptam_cg) has no FFI/plugin layer: its boundary for this path is the C++
 * class surface of KeyFrame / PatchFinder / Tracker / Bundle.  Each entry point below names the
 * reference interface it replaces (file:line, relative to the reference tree).  A header-only C++
 * shim that re-creates those class signatures over this ABI lives in ptam_shim.hpp.
 *
 * Conventions
 *   - every function returns an int status: PTAM_OK (0) or a negative PTAM_E_* code; algorithmic
 *     outcomes (found flags, accepted-iteration counts, the 32001 "out of border" ZMSSD sentinel,
 *     the -1 "bad template/level" sentinel) travel in out-parameters with the reference's values.
 *   - the caller keeps ownership of every host buffer; the library owns device memory behind
 *     opaque handles.  Bundle copies all inputs at add_* time like the reference
 *     (src/Bundle.cc:46-93).
 *   - a ptam_ctx is bound to ONE calling thread (own HIP stream + scratch).  The reference is
 *     entered from two threads (tracker: src/Tracker.cc:86, mapmaker: src/MapMaker.cc:57); give
 *     each its own ctx.  There is no global mutable state and no cached camera state
 *     (the reference's ATANCamera caches its last projection, include/ATANCamera.h:12-16).
 *   - poses are camera-from-world SE3 as 12 doubles: R row-major (9) then t (3).
 *   - images are 8-bit grey, row-major, `stride` bytes per row.
 *   - there is NO CPU fallback: if no gfx950 device / kernel image is available the create call
 *     fails with PTAM_E_HIP.
 */
#ifndef PTAM_HIP_H
#define PTAM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTAM_LEVELS 4                /* include/KeyFrame.h:34 */
#define PTAM_PATCH 8                 /* PatchFinder default nPatchSize, include/PatchFinder.h:55 */
#define PTAM_MAX_SSD (8 * 8 * 500)   /* src/PatchFinder.cc:18-19  -> 32000 */

enum {
    PTAM_OK = 0,
    PTAM_E_ARG = -1,      /* bad argument */
    PTAM_E_HIP = -2,      /* HIP runtime error (see ptam_last_error) */
    PTAM_E_STATE = -3,    /* call out of order */
    PTAM_E_LIMIT = -4,    /* size above a documented limit */
    PTAM_E_COMM = -5      /* collective failed */
};

/* libCVD halfSample rounding variant.  R = SSE2 pavgb/pavgw double rounding
 * (what an x86-64 libCVD build does for 16-byte aligned byte images), T = (a+b+c+d)/4 truncating. */
enum { PTAM_HALFSAMPLE_R = 0, PTAM_HALFSAMPLE_T = 1 };

/* cg::Tukey / Cauchy / Huber, include/Tools.h:128-228 */
enum { PTAM_EST_TUKEY = 0, PTAM_EST_CAUCHY = 1, PTAM_EST_HUBER = 2 };

typedef struct ptam_ctx ptam_ctx;
typedef struct ptam_kf ptam_kf;
typedef struct ptam_ba ptam_ba;

typedef struct { int32_t x, y; } ptam_int2;   /* CVD::ImageRef */

/* ATANCamera parameters: the 5-vector of config/camera.cfg:7 (normalised fx fy cx cy, w) and the
 * image size of ATANCamera::SetImageSize (src/ATANCamera.cc:21-25). */
typedef struct {
    double fx, fy, cx, cy, w;
    int32_t width, height;
} ptam_cam_params;

const char* ptam_last_error(void);           /* thread-local message of the last failure */
int ptam_device_count(int* n);

/* ---- context ------------------------------------------------------------------------------- */
/* replaces: ATANCamera(..) + RefreshParams (src/ATANCamera.cc:11-66); one per calling thread. */
int ptam_ctx_create(const ptam_cam_params* cam, int device, ptam_ctx** out);
int ptam_ctx_destroy(ptam_ctx* ctx);
int ptam_ctx_set_halfsample(ptam_ctx* ctx, int variant);
int ptam_ctx_sync(ptam_ctx* ctx);
void* ptam_ctx_stream(ptam_ctx* ctx);        /* the hipStream_t all of ctx's work is queued on */
/* derived camera constants (mdLargestRadius, mdMaxR, md2Tan, focal, centre: src/ATANCamera.cc:33-66) */
int ptam_ctx_camera_constants(ptam_ctx* ctx, double out[8]);

/* device memory helpers for callers that keep inputs resident (bench, shards) */
int ptam_dev_alloc(ptam_ctx* ctx, size_t bytes, void** dptr);
int ptam_dev_free(ptam_ctx* ctx, void* dptr);
int ptam_dev_upload(ptam_ctx* ctx, void* dptr, const void* host, size_t bytes);
int ptam_dev_download(ptam_ctx* ctx, void* host, const void* dptr, size_t bytes);

/* ---- KeyFrame::MakeKeyFrame_Lite  (src/KeyFrame.cc:18-54) ------------------------------------ */
int ptam_kf_create(ptam_ctx* ctx, int width, int height, ptam_kf** out);
int ptam_kf_destroy(ptam_kf* kf);
/* host image -> pyramid + FAST-10 (thresholds 10,15,15,10) + row LUTs, all on device. */
int ptam_make_keyframe_lite(ptam_ctx* ctx, ptam_kf* kf, const uint8_t* im, int stride);
/* same, image already resident on the device (stride == width) */
int ptam_make_keyframe_lite_dev(ptam_ctx* ctx, ptam_kf* kf, const uint8_t* d_im);
/* Level::operator= deep copy (include/KeyFrame.h:66-75) for the tracker->mapmaker hand-off
 * (src/MapMaker.cc:480-488), device to device. */
int ptam_kf_clone(ptam_ctx* ctx, const ptam_kf* src, ptam_kf** out);
int ptam_kf_level_info(ptam_ctx* ctx, const ptam_kf* kf, int level, int* w, int* h, int* n_corners);
/* host read-back of aLevels[level].im / vCorners / vCornerRowLUT; NULL pointers are skipped.
 * px: w*h bytes; corners: n_corners entries (raster order); rowlut: h ints. */
int ptam_kf_read_level(ptam_ctx* ctx, const ptam_kf* kf, int level,
                       uint8_t* px, ptam_int2* corners, int32_t* rowlut);

/* ---- KeyFrame::MakeKeyFrame_Rest (src/KeyFrame.cc:61-82) minus the SmallBlurryImage: libCVD fast_nonmax
 *      (score = largest threshold >= 10 at which the corner survives; kept iff no 8-neighbour corner
 *      scores strictly higher) and ImageProcess::ShiTomasiScoreAtPoint (src/ImageProcess.cc:20-47) on
 *      the maximal corners that lie >= 10 pixels inside the level.   */
int ptam_make_keyframe_rest(ptam_ctx* ctx, ptam_kf* kf);
int ptam_kf_rest_info(ptam_ctx* ctx, const ptam_kf* kf, int level, int* n_max_corners);
/* vMaxCorners (raster order) and, per maximal corner, its Shi-Tomasi score (-1.0 where the corner
 * is closer than 10 pixels to the border and the reference skips it).  Level::vCandidates is the
 * subset with score > MapMaker.CandidateMinShiTomasiScore (70; 400 in config/settings.cfg:27). */
int ptam_kf_read_rest(ptam_ctx* ctx, const ptam_kf* kf, int level, ptam_int2* max_corners, double* st_scores);

/* ---- PatchFinder::FindPatchCoarse / ZMSSDAtPoint  (src/PatchFinder.cc:160-211, 326-329;
 *      src/ImageProcess.cc:130-163) ------------------------------------------------------------- */
typedef struct {
    int32_t x, y;        /* irPos: predicted position, LEVEL-0 pixels (ir(v2Image), src/Tracker.cc:881) */
    int32_t level;       /* mnSearchLevel 0..3; < 0 = template bad, not searched */
    uint32_t range;      /* nRange in level-0 pixels */
} ptam_patch_query;

typedef struct {
    int32_t found;       /* mbFound */
    int32_t best_ssd;    /* nBestSSD; PTAM_MAX_SSD+1 when nothing scored */
    int32_t best_x, best_y;   /* irBest in search-level pixels (undefined when nothing scored: -1) */
    int32_t n_scored;    /* candidates that reached ZMSSDAtPoint */
    int32_t pad_;
    double pos[2];       /* mv2CoarsePos = LevelZeroPos(irBest, level), valid iff found */
} ptam_patch_result;

/* n queries, templates = n * 64 bytes (8x8 row-major, mimTemplate).  Host buffers. */
int ptam_find_patch_coarse_batch(ptam_ctx* ctx, const ptam_kf* kf, int n,
                                 const ptam_patch_query* queries, const uint8_t* templates,
                                 ptam_patch_result* results);
/* device-resident variant (queries/templates/results are device pointers), asynchronous */
int ptam_find_patch_coarse_batch_dev(ptam_ctx* ctx, const ptam_kf* kf, int n,
                                     const ptam_patch_query* d_queries, const uint8_t* d_templates,
                                     ptam_patch_result* d_results);
/* ZMSSDAtPoint of ONE template at n explicit points of a level (parity / epipolar use). */
int ptam_zmssd_at_points(ptam_ctx* ctx, const ptam_kf* kf, int level, int n,
                         const ptam_int2* points, const uint8_t* tmpl64, int32_t* ssd_out);

/* ---- PatchFinder::MakeSubPixTemplate + IterateSubPixToConvergence (src/PatchFinder.cc:219-318):
 *      inverse-compositional sub-pixel refinement of a coarse match (second half) */
typedef struct {
    double coarse_pos[2];   /* mv2CoarsePos, level-0 pixels (result of FindPatchCoarse) */
    int32_t level;          /* mnSearchLevel; < 0 = skip */
    int32_t max_its;        /* nMaxIts (8 in the tracker, src/Tracker.cc:583) */
} ptam_subpix_query;
typedef struct {
    int32_t converged;      /* return value of IterateSubPixToConvergence */
    int32_t iterations;     /* IterateSubPix calls made */
    double pos[2];          /* mv2SubPixPos */
    double mean_diff;       /* mdMeanDiff */
} ptam_subpix_result;
int ptam_subpix_batch(ptam_ctx* ctx, const ptam_kf* kf, int n, const ptam_subpix_query* queries,
                      const uint8_t* templates, ptam_subpix_result* results);

/* ---- PatchFinder::MakeTemplateCoarseCont (src/PatchFinder.cc:98-127): the 8x8 search template warped out
 *      of the map point's source keyframe by CVD::transform, then MakeTemplateSums.
 *      CVD::transform / CVD::sample are libCVD (not vendored by the reference): restated from its published
 *      source — positions advance by repeated addition (p += across per pixel, += down - 8*across per row),
 *      bilinear value (1-y)*((1-x)*a + x*b) + y*((1-x)*c + x*d) in double, converted to a byte by
 *      truncation; pixels whose source position leaves [0,w-1)x[0,h-1) become 0 and count as outside
 *      (checked per pixel only when the patch's bounding box is not wholly inside).
 *      The "same map point, warp moved < 0.07" reuse test (:103-111) is host state and stays with the
 *      caller (ptam::PatchFinder / host.PatchFinder keep mpLastTemplateMapPoint and mm2LastWarpMatrix). */
typedef struct {
    const ptam_kf* src_kf;      /* MapPoint::pPatchSourceKF */
    int32_t src_level;          /* MapPoint::nSourceLevel */
    int32_t search_level;       /* mnSearchLevel from CalcSearchLevelAndWarpMatrix; < 0 = template bad, skipped */
    int32_t center_x, center_y; /* MapPoint::irCenter, source-level pixels */
    double warp_inverse[4];     /* mm2WarpInverse row-major (ptam_pvs_result::warp_inverse) */
} ptam_template_query;
typedef struct {
    int32_t bad;                /* mbTemplateBad = (bool)nOutside (1 also when the query was skipped) */
    int32_t n_outside;          /* return value of CVD::transform */
    int32_t sum, sum_sq;        /* mnTemplateSum, mnTemplateSumSq */
    double m2[4];               /* the warp matrix used: M2Inverse(mm2WarpInverse) * LevelScale(mnSearchLevel) */
} ptam_template_result;
/* templates_out: n * 64 bytes (8x8 row-major), the layout ptam_find_patch_coarse_batch takes */
int ptam_make_templates_batch(ptam_ctx* ctx, int n, const ptam_template_query* queries, uint8_t* templates_out,
                              ptam_template_result* results);

/* ---- MapMaker::AddPointEpipolar: the corner scan (src/MapMaker.cc:598-637)   --------------
 * The line geometry (:541-596) stays with the caller — it is per-candidate scalar code on poses and depth
 * statistics; what moves to the device is the O(candidates x corners) part: MakeTemplateCoarseNoWarp of the
 * candidate in the source keyframe (:599, src/PatchFinder.cc:137-148), the in-plane corner table of the target
 * level (:604-614), the band / segment test of every target corner (:620-630) and ZMSSDAtPoint of the survivors
 * (:631-635), first strict minimum in corner order. */
/* Level::vImplaneCorners: UnProject(ir(LevelZeroPos(corner, level))) of every FAST corner of the level, built once
 * per (keyframe, level) and cached on the device (bImplaneCornersCached).  out_xy (nullable): room for cap corners,
 * 2 doubles each; n_out (nullable): the corner count. */
int ptam_kf_implane_corners(ptam_ctx* ctx, ptam_kf* kf, int level, double* out_xy, int cap, int* n_out);
typedef struct {
    int32_t level_x, level_y;   /* candidate.irLevelPos in the SOURCE keyframe, level pixels */
    double normal[2];           /* v2Normal */
    double norm_dist;           /* dNormDist */
    double along[2];            /* v2AlongProjectedLine */
    double min_len, max_len;    /* dMinLen, dMaxLen (after the +-2.0 clamps) */
    double max_dist_sq;         /* dMaxDistSq = (OnePixelDist() * (4 + nLevelScale))^2 */
} ptam_epipolar_query;
typedef struct {
    int32_t best;               /* nBest: index into the target level's vCorners, -1 = none */
    int32_t best_zmssd;         /* nBestZMSSD (PTAM_MAX_SSD + 1 when none) */
    int32_t n_scored;           /* corners that reached ZMSSDAtPoint */
    int32_t template_bad;       /* Finder.TemplateBad() after MakeTemplateCoarseNoWarp */
} ptam_epipolar_result;
int ptam_epipolar_search_batch(ptam_ctx* ctx, const ptam_kf* src, ptam_kf* target, int level, int n,
                               const ptam_epipolar_query* queries, ptam_epipolar_result* results);
/* ATANCamera::OnePixelDist() (src/ATANCamera.cc:69-75) for max_dist_sq */
int ptam_ctx_one_pixel_dist(ptam_ctx* ctx, double* out);
/* A deliberate deviation made observable.  TrackerData::ProjectAndDerivs (include/Tracker.h:89-94) calls GetProjectionDerivs()
 * after Project() even when Project() bailed out before the camera model (:73-80: a point behind the camera or outside the model's
 * radius) — the derivatives it then reads are the camera's cache of whichever point was projected last.  Here such a point keeps
 * its own derivatives of the previous iteration.  *out = how often that happened in the pose loops
 * (ptam_pose_gn*, ptam_track_map*, ptam_track_frame) on this context's device since the library was loaded: 0 means every
 * tracked frame so far was also bit-for-bit what the reference's data flow gives. */
int ptam_ctx_cache_hazards(ptam_ctx* ctx, long long* out);

/* ---- TrackerData::Project / ProjectAndDerivs + ATANCamera (include/Tracker.h:70-94,
 *      src/ATANCamera.cc:109-121, 179-209) ------------------------------------------------------- */
typedef struct {
    double cam[3];       /* v3Cam */
    double image[2];     /* v2Image */
    double derivs[4];    /* m2CamDerivs row-major */
    int32_t in_image;    /* bInImage */
    int32_t pad_;
} ptam_projection;
int ptam_project_points(ptam_ctx* ctx, int n, const double* world_xyz, const double pose[12],
                        ptam_projection* out);

/* TrackerData::Project (include/Tracker.h:70-85) on EXISTING TrackerData, as TrackMap re-projects the points it is about to
 * search after the coarse pose update (src/Tracker.cc:573-574, :606-608): cam always, image once the camera model is reached
 * (a point that bails out earlier keeps its previous v2Image), derivs untouched (ProjectAndDerivs refreshes them only for
 * found points, include/Tracker.h:89-94), in_image = bInImage. */
int ptam_reproject_points(ptam_ctx* ctx, int n, const double* world_xyz, const double pose[12], ptam_projection* inout);

/* ---- Tracker::TrackMap potentially-visible-set loop (src/Tracker.cc:453-478) with
 *      PatchFinder::CalcSearchLevelAndWarpMatrix (src/PatchFinder.cc:52-84)   */
typedef struct {
    double world[3];          /* MapPoint::v3WorldPos */
    double pixel_right_w[3];  /* MapPoint::v3PixelRight_W */
    double pixel_down_w[3];   /* MapPoint::v3PixelDown_W */
} ptam_pvs_point;
typedef struct {
    ptam_projection proj;     /* TData.v3Cam / v2Image / m2CamDerivs / bInImage */
    double warp_inverse[4];   /* mm2WarpInverse row-major (valid iff proj.in_image) */
    int32_t level;            /* nSearchLevel 0..3, or -1 (not in image, or inappropriate warp) */
    int32_t pad_;
} ptam_pvs_result;
/* counts (nullable): number of points per search level = sizes of avPVS[0..3] */
int ptam_track_pvs(ptam_ctx* ctx, int n, const ptam_pvs_point* points, const double pose[12],
                   ptam_pvs_result* results, int32_t counts[4]);

/* ---- MapMaker::ReFind_Common (src/MapMaker.cc:943-1020) for a batch of map points against ONE keyframe — the loop of
 *      ReFindInSingleKeyFrame (:1027-1042), and per keyframe of ReFindNewlyMade / ReFindFromFailureQueue  (SURVEY §8f rank 4).
 *      Per point: projection into the keyframe with the visibility tests of :950-975, camera derivatives,
 *      CalcSearchLevelAndWarpMatrix (its verdict is not looked at by the reference: the template is made at the level the
 *      loop stopped at and only CVD::transform's outside count decides TemplateBad, :979-986), FindPatchCoarse with range 4
 *      (:988), and for level > 0 the sub-pixel refinement whose convergence flag the reference ignores (:1000-1006).
 *      The set bookkeeping (sMeasurementKFs / sNeverRetryKFs tests of :947-948, inserts) stays with the caller. */
typedef struct {
    int32_t found;          /* return value: a measurement {level, root_pos, sub_pix, SRC_REFIND} is to be added */
    int32_t level;          /* m.nLevel = Finder.GetLevel(); -1 when the point never reached the PatchFinder */
    int32_t sub_pix;        /* m.bSubPix */
    int32_t never_retry;    /* the point-keyframe pair goes into sNeverRetryKFs (every failure path of the reference) */
    double root_pos[2];     /* m.v2RootPos, level-0 pixels (valid iff found) */
} ptam_refind_result;
int ptam_refind_batch(ptam_ctx* ctx, const ptam_kf* kf, const double kf_pose[12], int n, const ptam_pvs_point* points,
                      const ptam_template_query* sources /* src_kf, src_level, center_x / center_y */, ptam_refind_result* out);

/* ---- MapMaker::ReFind_Common as the reference CALLS it (src/MapMaker.cc:1046-1082): a list of (keyframe, point) pairs taken
 *      in order through ONE PatchFinder — `static PatchFinder Finder` (:977) — whose MakeTemplateCoarseCont keeps the
 *      template, its sums and mbTemplateBad when it last warped THIS map point and neither column of the warp has moved by
 *      more than 0.07 since (src/PatchFinder.cc:98-127).  ReFindNewlyMade (:1046-1066) is one new point against every
 *      keyframe in turn: consecutive keyframes with nearly the same pose search with the template of the first.  And a warp
 *      CalcSearchLevelAndWarpMatrix rejects sets mbTemplateBad (:78-81), which a kept template does not clear — that pair
 *      (and the kept ones after it) goes into sNeverRetryKFs although an earlier pair searched with the same template.
 *      The finder object carries that state from call to call like the static does; pairs with `skip` set (:947-948: the
 *      point is already measured in the keyframe, or never to be retried) return false without touching it. */
typedef struct ptam_refinder ptam_refinder;
int ptam_refinder_create(ptam_ctx* ctx, ptam_refinder** out);
int ptam_refinder_destroy(ptam_refinder* f);
typedef struct {
    const ptam_kf* kf;            /* k */
    double kf_pose[12];           /* k.se3CfromW */
    ptam_pvs_point point;         /* p: world position and pixel vectors */
    ptam_template_query source;   /* p: patch source (src_kf, src_level, center_x / center_y; the other fields are ignored) */
    int64_t point_id;             /* identity of the MapPoint (the reference compares &p with mpLastTemplateMapPoint) */
    int32_t skip;                 /* :947-948 */
    int32_t pad_;
} ptam_refind_pair;
/* template_kept (nullable): 1 where the finder searched with the template it already had */
int ptam_refind_pairs(ptam_ctx* ctx, ptam_refinder* finder, int n, const ptam_refind_pair* pairs, ptam_refind_result* out,
                      int32_t* template_kept);

/* ---- Tracker pose Gauss-Newton (src/Tracker.cc:613-643 driver, :928-1005 CalcPoseUpdate,
 *      include/Tracker.h:125-142 CalcJacobian/LinearUpdate) ---------------------------------------- */
typedef struct {
    double world[3];         /* MapPoint::v3WorldPos */
    double found[2];         /* v2Found (level-0 pixels) */
    double sqrt_inv_noise;   /* dSqrtInvNoise = 1 / 2^level (src/Tracker.cc:889) */
} ptam_pose_meas;

typedef struct {
    int32_t iterations;          /* 10 */
    uint32_t nonlinear_mask;     /* bit i set = full re-projection + Jacobian on iteration i;
                                    fine stage 0x211 (iter 0,4,9), coarse stage 0x3ff */
    int32_t override_after;      /* override sigma^2 used for iter > override_after (5) */
    double override_sigma_sq;    /* 16.0 fine, 1.0 coarse */
    int32_t mark_outliers_iter;  /* iteration whose weight==0 set is reported (9; -1 = none) */
    int32_t estimator;           /* PTAM_EST_* ("Tracker.MEstimator") */
    double prior;                /* wls.add_prior(100.0) */
} ptam_gn_opts;
void ptam_gn_opts_default(ptam_gn_opts* o);   /* fine-stage schedule of src/Tracker.cc:613-643 */

/* Runs the whole 10-iteration loop on the device in one launch.
 * pose_inout: mse3CamFromWorld.  entry (nullable): per-measurement state at loop entry as left by
 * the caller's last Project (used on iteration 0, which does not re-project, src/Tracker.cc:617);
 * NULL = project every point with the entry pose.  outlier_flags (nullable, n ints): 1 where
 * Weight()==0 on mark_outliers_iter.  updates_out (nullable): iterations*6 doubles, the v6Update
 * of every iteration. */
int ptam_pose_gn(ptam_ctx* ctx, int n, const ptam_pose_meas* meas, const ptam_projection* entry,
                 double pose_inout[12], const ptam_gn_opts* opts,
                 int32_t* outlier_flags, double* updates_out);

/* The same, also returning the measurements' TrackerData state when the loop ends (cam = v3Cam, image = v2Image, derivs =
 * m2CamDerivs as the last ProjectAndDerivs / LinearUpdate left them; in_image unspecified): what a FOLLOWING loop starts
 * from, because its iteration 0 does not re-project (src/Tracker.cc:617) — the coarse loop's points in the fine loop. */
int ptam_pose_gn_state(ptam_ctx* ctx, int n, const ptam_pose_meas* meas, const ptam_projection* entry,
                       double pose_inout[12], const ptam_gn_opts* opts, int32_t* outlier_flags, double* updates_out,
                       ptam_projection* state_out);

/* device-resident variant: every pointer is a device pointer (d_entry, d_outlier_flags, d_updates nullable; d_updates
 * holds 32*6 doubles), asynchronous on the context's stream — the measurements of a tracked frame are produced on
 * the device (patch search / sub-pixel results), so the frame needs no upload and only the 96-byte pose comes back.
 * n must be >= 1. */
int ptam_pose_gn_dev(ptam_ctx* ctx, int n, const ptam_pose_meas* d_meas, const ptam_projection* d_entry,
                     double* d_pose_inout, const ptam_gn_opts* opts, int32_t* d_outlier_flags, double* d_updates);

/* The same with the measurement count in device memory (d_n, clamped to [0, n_cap]; 0 = pose unchanged, zero updates,
 * src/Tracker.cc:955-956): the list ptam_gather_pose_meas_dev compacted is consumed without a host round trip.
 * Outlier flags are written for the first *d_n entries only.  pose_host_in (nullable, HOST pointer, 12 doubles): the entry
 * pose (the motion model's prediction) handed over as a kernel argument instead of being read from d_pose_inout — no
 * copy of its own.  pose_host_out (nullable, HOST pointer, 12 doubles): when
 * given, the call returns once the refined pose has arrived there — the kernel writes it into host-mapped memory, so a
 * tracked frame ends with one PCIe write instead of a D2H copy + stream synchronise; d_pose_inout is updated either way. */
int ptam_pose_gn_dev_counted(ptam_ctx* ctx, int n_cap, const int32_t* d_n, const ptam_pose_meas* d_meas,
                             const ptam_projection* d_entry, double* d_pose_inout, const ptam_gn_opts* opts,
                             int32_t* d_outlier_flags, double* d_updates, const double* pose_host_in, double* pose_host_out);

/* ---- Tracker::SearchForPoints' bookkeeping for a batch (src/Tracker.cc:883-909), device resident, asynchronous:
 *      query i with level >= 0 whose patch was found (d_results[i].found) — and, when d_subpix is given, whose
 *      sub-pixel refinement converged (:898-904) — becomes the next pose measurement, in query order:
 *      {v3WorldPos_i, v2Found = coarse position (:908) or sub-pixel position (:905), dSqrtInvNoise = 1 / LevelScale(level) (:889)}.
 *      d_world: the points' world positions, 3 doubles every world_stride_bytes (24 for packed xyz, sizeof(ptam_pvs_point) to
 *      read them out of the PVS input).  d_src_index (nullable): query index of every measurement (routes the outlier flags of
 *      the pose solve back to the map points).  d_count: number of measurements written.  d_level_found (nullable, 4 ints):
 *      manMeasFound per level (:892, :901). */
int ptam_gather_pose_meas_dev(ptam_ctx* ctx, int n, const ptam_patch_query* d_queries, const ptam_patch_result* d_results,
                              const ptam_subpix_result* d_subpix, const void* d_world, int world_stride_bytes,
                              ptam_pose_meas* d_meas_out, int32_t* d_src_index, int32_t* d_count, int32_t* d_level_found);

/* ---- Tracker::TrackMap (src/Tracker.cc:442-696) as ONE device-resident chain ------------------------------------------
 *      PVS loop -> choice of the coarse / top-level / fine search sets -> coarse SearchForPoints (range 30, sub-pixel) ->
 *      ten coarse pose iterations -> re-projection + SearchForPoints of the other sets -> ten fine pose iterations ->
 *      measurement and scene-depth bookkeeping.  List lengths, the mbDidCoarse decision and the fine search range are taken
 *      on the device; the call enqueues everything at once and returns when the result block has arrived (host-mapped
 *      memory, no copy).  The map (world positions, pixel vectors, patch sources) stays resident between frames, and so
 *      does every point's PatchFinder (TrackerData::Finder): MakeTemplateCoarseCont keeps a point's search template — and
 *      its mbTemplateBad — from frame to frame while neither column of the warp moves by more than 0.07
 *      (src/PatchFinder.cc:98-127), a warp rejected by CalcSearchLevelAndWarpMatrix leaves mbTemplateBad up until the next
 *      re-make (:78-81).  ptam_tracker_set_map starts every finder afresh. */
typedef struct ptam_tracker ptam_tracker;
typedef struct {
    int32_t try_coarse;         /* bTryCoarse after the caller's heuristics (src/Tracker.cc:505-516: DisableCoarse, velocity,
                                   just-recovered — then the caller also doubles coarse_max / coarse_range) */
    uint32_t coarse_min;        /* Tracker.CoarseMin 20        :492 */
    uint32_t coarse_max;        /* Tracker.CoarseMax 60        :493 */
    uint32_t coarse_range;      /* Tracker.CoarseRange 30      :494 */
    int32_t coarse_subpix_its;  /* Tracker.CoarseSubPixIts 8   :495 */
    int32_t max_patches;        /* Tracker.MaxPatchesPerFrame 1000  :593 */
    int32_t estimator;          /* Tracker.MEstimator */
    int32_t pad_;
} ptam_trackmap_opts;
void ptam_trackmap_opts_default(ptam_trackmap_opts* o);
typedef struct {
    double pose[12];            /* mse3CamFromWorld after the fine stage */
    int32_t did_coarse;         /* mbDidCoarse */
    int32_t n_pvs[4];           /* avPVS[l].size() after the PVS loop */
    int32_t attempted[4];       /* manMeasAttempted */
    int32_t found[4];           /* manMeasFound */
    int32_t n_coarse, n_top, n_fine;   /* sizes of the coarse set, of the remaining top-level set and of the fine set */
    int32_t n_meas;             /* found entries of vIterationSet = measurements of the fine pose loop */
    int32_t depth_n;            /* number of found points in the two sums below */
    int32_t templates_reused;   /* searched patches whose PatchFinder kept last frame's template (src/PatchFinder.cc:103-111) */
    int32_t pad_;
    double depth_sum, depth_sum_sq;   /* sums of v3Cam[2] and its square over the found points (:680-690) */
} ptam_trackmap_result;
/* one entry of vIterationSet, in its order (coarse set, top-level set, fine set) */
typedef struct {
    int32_t point;              /* map point index */
    int32_t level;              /* nSearchLevel (-1: bad template, not searched) */
    int32_t found;              /* bFound */
    int32_t did_subpix;         /* bDidSubPix */
    int32_t outlier;            /* Weight() == 0 on the fine loop's last iteration (:640, CalcPoseUpdate bMarkOutliers) */
    int32_t pad_;
    double v2_found[2];         /* v2Found, level-0 pixels (valid iff found) */
} ptam_trackmap_meas;
int ptam_tracker_create(ptam_ctx* ctx, int max_points, ptam_tracker** out);
int ptam_tracker_destroy(ptam_tracker* t);
/* the map: world position + pixel vectors per point, and its patch source (src_kf, src_level, center_x / center_y of
 * ptam_template_query; the other fields are ignored).  The source keyframes must stay alive while the tracker uses them. */
int ptam_tracker_set_map(ptam_tracker* t, int n, const ptam_pvs_point* points, const ptam_template_query* sources);
/* The map after the mapmaker changed it (points added at a new keyframe, bad points removed, positions refined by a bundle
 * adjustment): like ptam_tracker_set_map, but a point that was in the previous map keeps its TrackerData — prev_index[i] is the
 * index new point i had in the map of the last set_map / update_map call, -1 for a new point.  Its PatchFinder's template, warp
 * matrix and mbTemplateBad carry over as they do in the reference, where TrackerData lives as long as its MapPoint
 * (include/Tracker.h:42-67, src/Tracker.cc:453-464); new points start with a fresh finder.  No old index may appear twice. */
int ptam_tracker_update_map(ptam_tracker* t, int n, const ptam_pvs_point* points, const ptam_template_query* sources,
                            const int32_t* prev_index);
/* The frame's random orders, each a permutation of 0..n-1 (asynchronous upload): level l's PVS list is taken in the order
 * its members appear in shuffle_levels (replaces std::random_shuffle of avPVS[l], :483-484), the chop of the fine set to
 * MaxPatchesPerFrame in the order of shuffle_fine (:597-600).  Identity until set. */
int ptam_tracker_set_shuffle(ptam_tracker* t, const int32_t* shuffle_levels, const int32_t* shuffle_fine);
int ptam_track_map(ptam_tracker* t, const ptam_kf* current, const double pose_in[12], const ptam_trackmap_opts* opts,
                   ptam_trackmap_result* out);
/* A whole tracked frame in one call: KeyFrame::MakeKeyFrame_Lite (src/KeyFrame.cc:18-54) of the device-resident image
 * d_frame (stride == width) into `current`, then TrackMap against it: one entry, one queue, one wait. */
int ptam_track_map_frame(ptam_tracker* t, ptam_kf* current, const uint8_t* d_frame, const double pose_in[12],
                         const ptam_trackmap_opts* opts, ptam_trackmap_result* out);
/* ---- the tracker's motion model and Tracker::TrackFrame's tracking branch (src/Tracker.cc:134-137) ----------------------
 *      Host scalar code (no device work of its own): the decaying constant-velocity model of :1008-1056 and the bTryCoarse
 *      heuristics of :505-516, kept in a plain struct the caller owns.  The SmallBlurryImage rotation estimator
 *      (Tracker.UseRotationEstimator, :1017-1028) is outside this path (SmallBlurryImage is out of scope);
 *      this is the model with it switched off — the velocity comes from the last two tracked poses alone. */
typedef struct {
    double pose[12];               /* mse3CamFromWorld */
    double start_pose[12];         /* mse3StartPos: the pose before the prediction, :1015 */
    double velocity[6];            /* mv6CameraVelocity (translation, rotation) */
    double msd_scaled_velocity;    /* mdMSDScaledVelocityMagnitude, :1052-1055 */
    double scene_depth_mean;       /* mCurrentKF.dSceneDepthMean (1.0 after a reset, :56), updated when more than 20 points were found, :692-696 */
    double scene_depth_sigma;      /* mCurrentKF.dSceneDepthSigma */
    double coarse_min_velocity;    /* Tracker.CoarseMinVelocity 0.006, :496 */
    int32_t use_constant_velocity; /* Tracker.UseConstantVelocity 1, :1041 */
    int32_t disable_coarse;        /* Tracker.DisableCoarse 0, :495 */
    int32_t just_recovered;        /* mbJustRecoveredSoUseCoarse: the next frame tries the coarse stage with doubled CoarseMax / CoarseRange, :508-513 */
    int32_t pad_;
} ptam_motion_model;
/* Tracker::Reset's part of the model (:52-56): pose = the given one, zero velocity, depth mean 1, default tunables */
void ptam_motion_reset(ptam_motion_model* m, const double pose[12]);
/* Tracker::PredictPoseWithMotionModel :1013-1030: start_pose = pose; pose = exp(velocity) * start_pose */
void ptam_motion_predict(ptam_motion_model* m);
/* the end of TrackMap (:692-696: scene depth from the frame's sums when depth_n > 20) + Tracker::UpdateMotionModel :1036-1056 with
 * pose = r->pose: velocity = ln(pose * start_pose^-1) (or its decaying mix), msd_scaled_velocity = |(v_t / depth mean, v_w)| */
void ptam_motion_update(ptam_motion_model* m, const ptam_trackmap_result* r);
/* TooN SE3<>::exp / SE3<>::ln on (R row-major | t) poses; mu = (translation part, rotation vector) */
void ptam_se3_exp(const double mu[6], double pose_out[12]);
void ptam_se3_ln(const double pose[12], double mu_out[6]);
/* One tracked frame of a camera that moves (src/Tracker.cc:134-137 with :94 before them): MakeKeyFrame_Lite of the
 * device-resident frame into `current`, PredictPoseWithMotionModel, the bTryCoarse heuristics (*opts is the caller's tunables;
 * its try_coarse is ignored and decided here, coarse_max / coarse_range doubled after a recovery), TrackMap from the predicted
 * pose, UpdateMotionModel.  m->pose is the tracked pose afterwards.  A call that returns an error leaves *m untouched (the
 * model is advanced on a copy and committed on success): the frame can be retried. */
int ptam_track_frame(ptam_tracker* t, ptam_kf* current, const uint8_t* d_frame, ptam_motion_model* m,
                     const ptam_trackmap_opts* opts, ptam_trackmap_result* out);

/* nb frames of nb independent trackers (own context, map, keyframe, prediction each) as ONE chain of launches on the first
 * tracker's queue: every kernel of the chain gets a second grid dimension, row i works on frame i.  Not part of the
 * reference's surface (it tracks one camera); it exists because a process gets four hardware queues and a tracked frame
 * occupies one for its whole dependent chain, so separate calls keep at most four frames in flight whatever the device has
 * idle.  Frame i's result is what ptam_track_map_frame(trackers[i], current[i], d_frames[i], poses_in + 12 i, opts) gives
 * (bit for bit when the batch's list capacities select the same pose-kernel instantiation as the single call: maps of equal
 * size do).  The trackers must sit on one device in DIFFERENT contexts with the same camera model, image size and halfSample
 * variant; set the frames' permutations beforehand (ptam_tracker_set_shuffle). */
int ptam_track_map_frames_batch(int nb, ptam_tracker* const* trackers, ptam_kf* const* current, const uint8_t* const* d_frames,
                                const double* poses_in, const ptam_trackmap_opts* opts, ptam_trackmap_result* out);
/* vIterationSet of the last frame (what :667-676 turns into mCurrentKF.mMeasurements): *n = its length; out (nullable)
 * receives up to cap entries. */
int ptam_tracker_read_iteration_set(ptam_tracker* t, ptam_trackmap_meas* out, int cap, int* n);

/* One Tracker::CalcPoseUpdate (src/Tracker.cc:928-1005) on caller-provided Jacobians. */
typedef struct {
    double found[2];
    double image[2];
    double sqrt_inv_noise;
    double jac[12];          /* m26Jacobian row-major 2x6 */
} ptam_pose_update_meas;
int ptam_calc_pose_update(ptam_ctx* ctx, int n, const ptam_pose_update_meas* meas,
                          double override_sigma_sq, int estimator, double prior,
                          double mu_out[6], int32_t* weight_zero_flags);

/* ---- Bundle (include/Bundle.h:106-152, src/Bundle.cc) -------------------------------------------- */
typedef struct {
    int32_t max_iterations;          /* Bundle.MaxIterations = 20          src/Bundle.cc:40 */
    double update_sq_conv_limit;     /* UpdateSquaredConvergenceLimit 1e-6  src/Bundle.cc:41 */
    double min_sigma;                /* Bundle.MinTukeySigma = 0.4          src/Bundle.cc:234 */
    int32_t estimator;               /* Bundle.MEstimator = Tukey           src/Bundle.cc:132 */
    int32_t verbose;                 /* Bundle.Cout                         src/Bundle.cc:42 */
    int32_t deterministic;           /* not in the reference: 1 = camera sums of pass 2 (src/Bundle.cc:315-321) in a fixed order —
                                        the accumulation kernel stores each measurement's weighted A (2x6) and epsilon, a
                                        camera-major pass adds them up; two runs of the same problem are then bit-identical.
                                        0 (default): LDS atomics, whose order differs from run to run in the last bits */
    int32_t pad_;
} ptam_ba_opts;
void ptam_ba_opts_default(ptam_ba_opts* o);

/* one lambda trial (the unit mnCounter counts, src/Bundle.cc:518); mirrors the cout line :508 */
typedef struct {
    double lambda;          /* mdLambda used by this trial */
    double sigma_sq;        /* mdSigmaSquared of the enclosing LM step */
    double err_old;         /* dCurrentError */
    double err_new;         /* dNewError */
    double sum_sq_update;   /* |da|^2 + |db|^2 */
    int32_t n_bad;          /* measurements flagged bBad in the enclosing step (so far) */
    int32_t accepted;       /* 1 if this trial's step was committed */
} ptam_ba_trial;

int ptam_ba_create(ptam_ctx* ctx, const ptam_ba_opts* opts, ptam_ba** out);
int ptam_ba_destroy(ptam_ba* ba);
/* Bundle::AddCamera src/Bundle.cc:46-63: returns the camera id (>= 0) */
int ptam_ba_add_camera(ptam_ba* ba, const double pose[12], int fixed);
/* Bundle::AddPoint src/Bundle.cc:66-79 (NaN point -> zeros): returns the point id (>= 0) */
int ptam_ba_add_point(ptam_ba* ba, const double pos[3]);
/* Bundle::AddMeas src/Bundle.cc:82-93; sigma_sq = 4^level (src/MapMaker.cc:879-880) */
int ptam_ba_add_meas(ptam_ba* ba, int cam, int point, const double found[2], double sigma_sq);
/* bulk marshalling of the same three calls (ids are assigned in array order) */
int ptam_ba_add_cameras(ptam_ba* ba, int n, const double* poses12, const uint8_t* fixed);
int ptam_ba_add_points(ptam_ba* ba, int n, const double* pos3);
int ptam_ba_add_measurements(ptam_ba* ba, int n, const int32_t* cam, const int32_t* point,
                             const double* found2, const double* sigma_sq);
/* Bundle::Compute src/Bundle.cc:116-158.  *abort_flag (nullable) is polled on the host between
 * device trials (src/Bundle.cc:134,338); a sharded bundle sums the flags of all ranks with every trial's
 * scalars and acts on the sum, so that all ranks stop at the same trial (one trial later than a local
 * test would).  *accepted_out = mnAccepted, always >= 0: the reference's `return -1` (src/Bundle.cc:149-150)
 * is dead code there — Do_LM_Step returns true unconditionally (:550), TooN's LDL^T signals nothing — and
 * a singular or non-finite system shows up as it does in the reference: NaN errors, rejected trials, and
 * (where the reference would loop for ever, :118-123) an early stop.  Errors of the call itself (< 0
 * status: PTAM_E_ARG for a duplicate measurement, HIP / communicator failures) are the return value, not
 * *accepted_out.  Like src/Bundle.cc:46-93 the device path bounds neither the cameras of a bundle nor the measurements of a
 * point (a point seen by more than 256 keyframes and more than ~600 free cameras take slower kernel forms). */
int ptam_ba_compute(ptam_ba* ba, const volatile unsigned char* abort_flag, int* accepted_out);
int ptam_ba_converged(const ptam_ba* ba);                    /* include/Bundle.h:115 */
int ptam_ba_get_point(const ptam_ba* ba, int n, double pos[3]);      /* src/Bundle.cc:613 */
int ptam_ba_get_camera(const ptam_ba* ba, int n, double pose[12]);   /* src/Bundle.cc:618 */
int ptam_ba_get_all(const ptam_ba* ba, double* poses12, double* points3);
/* GetOutlierMeasurements src/Bundle.cc:623: (point, camera) pairs in the reference's order
 * (LM step, then measurement insertion order).  Returns the count; writes up to cap pairs. */
int ptam_ba_get_outliers(const ptam_ba* ba, int32_t* point_cam_pairs, int cap);
int ptam_ba_get_trials(const ptam_ba* ba, ptam_ba_trial* out, int cap);   /* returns count */
int ptam_ba_counts(const ptam_ba* ba, int* n_cams, int* n_free_cams, int* n_points, int* n_meas);
/* Trials of this bundle that were run again with the launch-per-block-column form of the camera solve because the persistent
 * form (one launch whose workgroups hand tiles to each other through flags) gave up a wait — it needs all of its workgroups
 * resident, which a device shared with another process' solve may not grant.  The repeated trial's result is what the reference
 * computes; the rest of the adjustment keeps the slower form.  0 in normal operation (returned as the function's value). */
int ptam_ba_solve_fallbacks(const ptam_ba* ba);
/* The other deliberate deviation made observable.  The reference accepts a second measurement of a point by the same camera
 * (both stay in mMeasList, src/Bundle.cc:76-93, while the look-up table of :558-567 keeps the last one: the two are summed into
 * U, V and the gradients but only one reaches the Schur complement); here ptam_ba_prepare / ptam_ba_compute refuse such a bundle
 * with PTAM_E_ARG.  Returns how many measurements of the last prepare had a twin (0 for a bundle that was accepted). */
int ptam_ba_duplicates_refused(const ptam_ba* ba);
/* Operating switches of the camera solve, read from the environment once per process: PTAM_LDLT_NO_CHAIN=1 uses the
 * launch-per-block-column form everywhere (a device shared between processes that all adjust bundles); PTAM_CH_SPIN_LIMIT=<n> is
 * the number of looks (~1 us each, default 2^18) a workgroup of the persistent form takes before it gives up a wait.
 * PTAM_TWO_QUEUES=1 sends a rejected trial's continuation to the context's second queue, which is empty, instead of behind the
 * kernels that were enqueued for the other outcome (~6 us per rejected trial); the trial's decision is then a launch of its own
 * again (2.3 us per trial: the default folds it into the next step's first launch, which is only safe on one queue). */

/* profiling hooks used by bench.py: HIP-event timing of individual kernels on the ctx stream. */
enum {
    PTAM_K_PROJECT = 0,     /* K5 pass-1 project + e^2 */
    PTAM_K_SELECT = 1,      /* K6 order statistic */
    PTAM_K_JACOBIAN = 2,    /* K7 fused Jacobian + normal-equation accumulate (roofline kernel) */
    PTAM_K_VINV = 3,
    PTAM_K_SCHUR = 4,       /* K8 */
    PTAM_K_SOLVE = 5,       /* K9 */
    PTAM_K_UPDATE = 6,      /* K10 back-substitution + apply + new error */
    PTAM_K_EXCHANGE = 7,    /* sharded bundles only: the all-reduce of S|E (the path's one exchange step) */
    PTAM_K_COUNT = 8
};
int ptam_ba_set_profiling(ptam_ba* ba, int on);
int ptam_ba_kernel_time(const ptam_ba* ba, int kernel, double* total_ms, int* launches);
/* upload + build the device structures without running (Compute does this implicitly) */
int ptam_ba_prepare(ptam_ba* ba);
/* (measurement-only entry points — the K7 launch bracket, the native frame drivers — are declared in ptam_hip_bench.h) */

/* ---- sharded global BA: measurements sharded by point across ranks ----------- */
/* A collective hook: all-reduce (sum) `count` doubles in place at device pointer `dptr`,
 * ordered on `stream` (hipStream_t).  Return 0 on success. */
typedef int (*ptam_allreduce_f64_fn)(void* user, double* dptr, size_t count, void* stream);
/* Attach a communicator to a bundle: `rank`'s bundle holds ALL cameras but only its shard of the
 * points (and every measurement of those points).  With a hook attached, Compute all-reduces the
 * camera system (S lower blocks + E), the error scalars and the median histogram per trial.
 * Failure semantics: what decides the SEQUENCE of collectives is itself collective — the abort flag, a shard that cannot
 * be prepared, a rank without measurements — so every rank leaves Compute at the same trial with an error or a result.
 * A failure that strikes ONE rank between two collectives of a trial (a HIP error, an allocation failure, the hook
 * returning non-zero) is returned by that rank at once; the others are by then inside (or on their way into) the next
 * all-reduce and can only be released by the communicator: the hook MUST time out or be abortable (RCCL: the
 * communicator's own abort / watchdog; torch.distributed: the process group's timeout), after which the process group is
 * to be torn down — the bundle's device state is undefined. */
int ptam_ba_set_comm(ptam_ba* ba, int rank, int world, ptam_allreduce_f64_fn fn, void* user);

/* built-in RCCL implementation of the hook (librccl is dlopen'ed on first use) */
typedef struct ptam_rccl ptam_rccl;
int ptam_rccl_unique_id(uint8_t id_out[128]);
int ptam_rccl_create(ptam_ctx* ctx, const uint8_t id[128], int rank, int world, ptam_rccl** out);
int ptam_rccl_destroy(ptam_rccl* c);
int ptam_rccl_allreduce_f64(void* comm /* ptam_rccl* */, double* dptr, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PTAM_HIP_H */
