// ptam_shim.hpp — header-only C++ shim that re-creates the reference's class surface for the hot
// path on top of the C ABI of ptam_hip.h, so the two-thread SLAM loop of cggos/ptam_cg can switch
// KeyFrame::MakeKeyFrame_Lite / PatchFinder::FindPatchCoarse / Tracker::CalcPoseUpdate + pose loop /
// Bundle to the MI355X path by changing includes (INTEGRATION.md shows the edits).
//
// TooN / libCVD are not available to this repo, so the shim uses POD stand-ins with the same
// meaning: ptam::SE3 (R row-major + t) for TooN::SE3<>, ptam::ImageRef for CVD::ImageRef,
// ptam::Vec<N> for TooN::Vector<N>.  A maintainer with TooN at hand converts with two memcpy's
// (see INTEGRATION.md §3).  Each method cites the reference declaration it mirrors.
#ifndef PTAM_SHIM_HPP
#define PTAM_SHIM_HPP

#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ptam_hip.h"

namespace ptam {

static_assert(sizeof(bool) == 1, "abort flag is passed as one byte");

struct ImageRef {   // CVD::ImageRef
    int x, y;
};
template <int N>
using Vec = std::array<double, N>;   // TooN::Vector<N>

struct SE3 {   // TooN::SE3<> : camera-from-world
    double R[9];   // row-major
    double t[3];
    static SE3 Identity() {
        SE3 s{};
        s.R[0] = s.R[4] = s.R[8] = 1.0;
        return s;
    }
    void to12(double* p) const {
        std::memcpy(p, R, sizeof R);
        std::memcpy(p + 9, t, sizeof t);
    }
    static SE3 from12(const double* p) {
        SE3 s;
        std::memcpy(s.R, p, sizeof s.R);
        std::memcpy(s.t, p + 9, sizeof s.t);
        return s;
    }
};

inline void check(int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + ptam_last_error());
}

// One per calling thread (tracker thread / mapmaker thread): replaces the per-owner ATANCamera copies
// of the reference (include/MapMaker.h:61, include/Tracker.h) and owns the HIP stream.
class Context {
public:
    // vParams = Camera.Parameters (config/camera.cfg:7), irSize = ATANCamera::SetImageSize
    Context(const Vec<5>& vParams, ImageRef irSize, int device = 0) {
        ptam_cam_params p{vParams[0], vParams[1], vParams[2], vParams[3], vParams[4], irSize.x, irSize.y};
        check(ptam_ctx_create(&p, device, &h_), "ptam_ctx_create");
        size_ = irSize;
    }
    ~Context() { ptam_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ptam_ctx* handle() const { return h_; }
    ImageRef size() const { return size_; }
    void SetHalfSampleVariant(int v) { check(ptam_ctx_set_halfsample(h_, v), "set_halfsample"); }

private:
    ptam_ctx* h_ = nullptr;
    ImageRef size_{0, 0};
};

// struct Level (include/KeyFrame.h:55-124): host views are fetched lazily from the device
struct Level {
    int w = 0, h = 0;
    std::vector<uint8_t> im;                 // CVD::Image<CVD::byte>
    std::vector<ImageRef> vCorners;          // FAST corners, raster order
    std::vector<int> vCornerRowLUT;
    static int LevelScale(int nLevel) { return 1 << nLevel; }                                   // :85-88
    static double LevelZeroPos(double dLevelPos, int nLevel) { return (dLevelPos + 0.5) * LevelScale(nLevel) - 0.5; }
    static double LevelNPos(double dRootPos, int nLevel) { return (dRootPos + 0.5) / LevelScale(nLevel) - 0.5; }
};

// struct KeyFrame (include/KeyFrame.h:130-149)
class KeyFrame {
public:
    explicit KeyFrame(Context& c) : ctx_(&c) { check(ptam_kf_create(c.handle(), c.size().x, c.size().y, &h_), "ptam_kf_create"); }
    ~KeyFrame() { ptam_kf_destroy(h_); }
    KeyFrame(const KeyFrame& o) : ctx_(o.ctx_) {   // Level::operator= deep copy (:66-75), device side
        check(ptam_kf_clone(ctx_->handle(), o.h_, &h_), "ptam_kf_clone");
    }
    KeyFrame& operator=(const KeyFrame& o) {
        if (this != &o) {
            ptam_kf* n = nullptr;
            check(ptam_kf_clone(o.ctx_->handle(), o.h_, &n), "ptam_kf_clone");
            ptam_kf_destroy(h_);
            h_ = n;
            ctx_ = o.ctx_;
            fetched_ = 0;
        }
        return *this;
    }
    // void MakeKeyFrame_Lite(CVD::BasicImage<CVD::byte>& im)   include/KeyFrame.h:141, src/KeyFrame.cc:18
    void MakeKeyFrame_Lite(const uint8_t* im, int stride) {
        check(ptam_make_keyframe_lite(ctx_->handle(), h_, im, stride), "ptam_make_keyframe_lite");
        fetched_ = 0;
    }
    // aLevels[l] host view (pixels, vCorners, vCornerRowLUT)
    const Level& aLevels(int l) {
        if (!(fetched_ & (1u << l))) {
            Level& L = lev_[l];
            int n = 0;
            check(ptam_kf_level_info(ctx_->handle(), h_, l, &L.w, &L.h, &n), "ptam_kf_level_info");
            L.im.resize((size_t)L.w * L.h);
            L.vCorners.resize(n);
            L.vCornerRowLUT.resize(L.h);
            static_assert(sizeof(ImageRef) == sizeof(ptam_int2), "layout");
            check(ptam_kf_read_level(ctx_->handle(), h_, l, L.im.data(), reinterpret_cast<ptam_int2*>(L.vCorners.data()),
                                     L.vCornerRowLUT.data()),
                  "ptam_kf_read_level");
            fetched_ |= 1u << l;
        }
        return lev_[l];
    }
    ptam_kf* handle() const { return h_; }
    SE3 se3CfromW = SE3::Identity();
    bool bFixed = false;

private:
    Context* ctx_;
    ptam_kf* h_ = nullptr;
    Level lev_[PTAM_LEVELS];
    unsigned fetched_ = 0;
};

// class PatchFinder (include/PatchFinder.h:51-137): coarse search stage
class PatchFinder {
public:
    explicit PatchFinder(Context& c) : ctx_(&c) {}
    // template + level as left by MakeTemplateCoarseCont / NoWarp (src/PatchFinder.cc:98-148)
    void SetTemplate(const uint8_t tmpl64[64], int nSearchLevel, bool bBad = false) {
        std::memcpy(tmpl_, tmpl64, 64);
        mnSearchLevel = nSearchLevel;
        mbTemplateBad = bBad;
    }
    // void MakeTemplateCoarseCont(MapPoint& p)  include/PatchFinder.h:70, src/PatchFinder.cc:98-127.
    // The MapPoint fields it reads are passed explicitly: pPatchSourceKF, nSourceLevel, irCenter; mm2WarpInverse and
    // mnSearchLevel are what CalcSearchLevelAndWarpMatrix left (TrackMapPVS returns both per point).  `pPointId`
    // stands for the MapPoint's address in the reuse test ("same point and the warp moved < 0.07": keep the template).
    void MakeTemplateCoarseCont(const void* pPointId, KeyFrame& kfSource, int nSourceLevel, ImageRef irCenter,
                                const double mm2WarpInverse[4], int nSearchLevel) {
        mnSearchLevel = nSearchLevel;
        if (nSearchLevel < 0) {   // CalcSearchLevelAndWarpMatrix returned -1 and set mbTemplateBad (src/PatchFinder.cc:78-81)
            mbTemplateBad = true;
            return;
        }
        // m2 = M2Inverse(mm2WarpInverse) * LevelScale(mnSearchLevel)   (include/Tools.h:54-65)
        const double det = mm2WarpInverse[0] * mm2WarpInverse[3] - mm2WarpInverse[2] * mm2WarpInverse[1];
        const double inv = 1.0 / det, sc = (double)(1 << nSearchLevel);
        const double m2[4] = {mm2WarpInverse[3] * inv * sc, -mm2WarpInverse[1] * inv * sc, -mm2WarpInverse[2] * inv * sc,
                              mm2WarpInverse[0] * inv * sc};
        bool refresh = pPointId != mpLastTemplateMapPoint;
        for (int i = 0; !refresh && i < 2; i++) {   // columns of m2 against the last warp matrix
            const double d0 = m2[i] - mm2LastWarpMatrix[i], d1 = m2[2 + i] - mm2LastWarpMatrix[2 + i];
            if (d0 * d0 + d1 * d1 > 0.07 * 0.07) refresh = true;
        }
        if (!refresh) return;
        ptam_template_query q{kfSource.handle(), nSourceLevel, nSearchLevel, irCenter.x, irCenter.y,
                              {mm2WarpInverse[0], mm2WarpInverse[1], mm2WarpInverse[2], mm2WarpInverse[3]}};
        ptam_template_result r;
        check(ptam_make_templates_batch(ctx_->handle(), 1, &q, tmpl_, &r), "ptam_make_templates_batch");
        mbTemplateBad = r.bad != 0;
        mpLastTemplateMapPoint = pPointId;
        for (int i = 0; i < 4; i++) mm2LastWarpMatrix[i] = r.m2[i];
    }
    // the batched form for SearchForPoints: every template of a frame in one launch (no reuse test: callers that want
    // it keep (point id, m2) per point and drop the unchanged queries before the call)
    static void MakeTemplatesBatch(Context& c, const std::vector<ptam_template_query>& q, std::vector<uint8_t>& templates64,
                                   std::vector<ptam_template_result>& out) {
        templates64.resize(q.size() * 64);
        out.resize(q.size());
        check(ptam_make_templates_batch(c.handle(), (int)q.size(), q.data(), templates64.data(), out.data()),
              "ptam_make_templates_batch");
    }
    // bool FindPatchCoarse(CVD::ImageRef ir, KeyFrame& kf, unsigned int nRange)  include/PatchFinder.h:78
    bool FindPatchCoarse(ImageRef irPos, KeyFrame& kf, unsigned int nRange) {
        ptam_patch_query q{irPos.x, irPos.y, mbTemplateBad ? -1 : mnSearchLevel, nRange};
        ptam_patch_result r;
        check(ptam_find_patch_coarse_batch(ctx_->handle(), kf.handle(), 1, &q, tmpl_, &r), "ptam_find_patch_coarse_batch");
        mbFound = r.found != 0;
        if (mbFound) mv2CoarsePos = {r.pos[0], r.pos[1]};
        return mbFound;
    }
    // the batched form SearchForPoints (src/Tracker.cc:867-912) should use: one launch for all patches
    static void FindPatchCoarseBatch(Context& c, KeyFrame& kf, const std::vector<ptam_patch_query>& q,
                                     const std::vector<uint8_t>& templates64, std::vector<ptam_patch_result>& out) {
        out.resize(q.size());
        check(ptam_find_patch_coarse_batch(c.handle(), kf.handle(), (int)q.size(), q.data(), templates64.data(), out.data()),
              "ptam_find_patch_coarse_batch");
    }
    // the corner scan of MapMaker::AddPointEpipolar (src/MapMaker.cc:598-637) for every candidate of a source
    // keyframe level at once: MakeTemplateCoarseNoWarp + band / segment test over kTarget's in-plane corners +
    // ZMSSDAtPoint; result[i].best indexes kTarget.aLevels(nLevel).vCorners.  The caller keeps the line geometry
    // (:541-596) and fills one ptam_epipolar_query per candidate; max_dist_sq = (OnePixelDist * (4 + nLevelScale))^2.
    static void EpipolarSearchBatch(Context& c, KeyFrame& kSrc, KeyFrame& kTarget, int nLevel,
                                    const std::vector<ptam_epipolar_query>& q, std::vector<ptam_epipolar_result>& out) {
        out.resize(q.size());
        check(ptam_epipolar_search_batch(c.handle(), kSrc.handle(), kTarget.handle(), nLevel, (int)q.size(), q.data(), out.data()),
              "ptam_epipolar_search_batch");
    }
    // int ZMSSDAtPoint(CVD::BasicImage<CVD::byte>&, const CVD::ImageRef&)   include/PatchFinder.h:79
    int ZMSSDAtPoint(KeyFrame& kf, int nLevel, ImageRef ir) {
        ptam_int2 p{ir.x, ir.y};
        int32_t v = 0;
        check(ptam_zmssd_at_points(ctx_->handle(), kf.handle(), nLevel, 1, &p, tmpl_, &v), "ptam_zmssd_at_points");
        return v;
    }
    // void MakeSubPixTemplate(); bool IterateSubPixToConvergence(KeyFrame&, int nMaxIts)
    //   include/PatchFinder.h:83-87, src/PatchFinder.cc:219-318 — starts at the coarse position
    bool IterateSubPixToConvergence(KeyFrame& kf, int nMaxIts) {
        ptam_subpix_query q{{mv2CoarsePos[0], mv2CoarsePos[1]}, mnSearchLevel, nMaxIts};
        ptam_subpix_result r;
        check(ptam_subpix_batch(ctx_->handle(), kf.handle(), 1, &q, tmpl_, &r), "ptam_subpix_batch");
        mv2SubPixPos = {r.pos[0], r.pos[1]};
        return r.converged != 0;
    }
    Vec<2> GetSubPixPos() const { return mv2SubPixPos; }
    Vec<2> GetCoarsePosAsVector() const { return mv2CoarsePos; }
    int GetLevel() const { return mnSearchLevel; }
    bool TemplateBad() const { return mbTemplateBad; }

private:
    Context* ctx_;
    uint8_t tmpl_[64] = {0};
    int mnSearchLevel = 0;
    bool mbTemplateBad = false, mbFound = false;
    const void* mpLastTemplateMapPoint = nullptr;            // include/PatchFinder.h:135-136
    double mm2LastWarpMatrix[4] = {9999.9, 0, 0, 9999.9};    // src/PatchFinder.cc:21-22
    Vec<2> mv2CoarsePos{0, 0}, mv2SubPixPos{0, 0};
};

// TrackMap's potentially-visible-set loop (src/Tracker.cc:453-478): TData.Project + GetProjectionDerivs
// + Finder.CalcSearchLevelAndWarpMatrix for every map point, one launch.
inline void TrackMapPVS(Context& c, const std::vector<ptam_pvs_point>& vMapPoints, const SE3& se3CamFromWorld,
                        std::vector<ptam_pvs_result>& out, int anPVSSize[PTAM_LEVELS] = nullptr) {
    double pose[12];
    se3CamFromWorld.to12(pose);
    out.resize(vMapPoints.size());
    int32_t counts[4];
    check(ptam_track_pvs(c.handle(), (int)vMapPoints.size(), vMapPoints.data(), pose, out.data(), counts), "ptam_track_pvs");
    if (anPVSSize)
        for (int l = 0; l < PTAM_LEVELS; l++) anPVSSize[l] = counts[l];
}

// The part of TrackerData (include/Tracker.h:41-145) the pose loop reads
struct TrackerDataLite {
    Vec<3> v3WorldPos;       // Point.v3WorldPos
    Vec<2> v2Found;          // v2Found (L0 pixels)
    double dSqrtInvNoise;    // 1 / 2^level
    bool bOutlier = false;   // set where Tracker::CalcPoseUpdate would ++nMEstimatorOutlierCount
};

// Tracker::TrackMap's ten Gauss-Newton pose iterations (src/Tracker.cc:613-643; :552-568 when bCoarse),
// including every CalcPoseUpdate (:928-1005), in one device launch.
inline SE3 TrackMapPoseIterations(Context& c, std::vector<TrackerDataLite>& vTD, const SE3& se3CamFromWorld,
                                  bool bCoarse = false, int nEstimator = PTAM_EST_TUKEY) {
    std::vector<ptam_pose_meas> m(vTD.size());
    for (size_t i = 0; i < vTD.size(); i++) {
        std::memcpy(m[i].world, vTD[i].v3WorldPos.data(), 24);
        std::memcpy(m[i].found, vTD[i].v2Found.data(), 16);
        m[i].sqrt_inv_noise = vTD[i].dSqrtInvNoise;
    }
    ptam_gn_opts o;
    ptam_gn_opts_default(&o);
    o.estimator = nEstimator;
    if (bCoarse) {
        o.nonlinear_mask = 0x3ff;
        o.override_sigma_sq = 1.0;
        o.mark_outliers_iter = -1;
    }
    double pose[12];
    se3CamFromWorld.to12(pose);
    std::vector<int32_t> flags(vTD.size());
    check(ptam_pose_gn(c.handle(), (int)m.size(), m.data(), nullptr, pose, &o, flags.data(), nullptr), "ptam_pose_gn");
    for (size_t i = 0; i < vTD.size(); i++) vTD[i].bOutlier = flags[i] != 0;
    return SE3::from12(pose);
}

// The fine stage of a tracked frame with nothing but the pose crossing PCIe: SearchForPoints over the potentially
// visible set (src/Tracker.cc:867-912: FindPatchCoarse per point, every found patch becomes a measurement with
// v2Found = coarse position and dSqrtInvNoise = 1 / LevelScale) followed by the ten pose iterations (:613-643).
// Queries / templates / world positions are device buffers the caller filled (ptam_dev_upload, or the outputs of
// ptam_track_pvs / ptam_make_templates_batch kept resident); results stay resident for the caller to read lazily
// (outlier flags + source indices route nMEstimatorOutlierCount back to the map points).
class ResidentFrameTracker {
public:
    ResidentFrameTracker(Context& c, int nCapacity) : c_(c), cap_(nCapacity) {
        alloc(&res_, sizeof(ptam_patch_result) * (size_t)cap_);
        alloc(&meas_, sizeof(ptam_pose_meas) * (size_t)cap_);
        alloc(&src_, 4 * (size_t)cap_);
        alloc(&flags_, 4 * (size_t)cap_);
        alloc(&count_, 32);
        alloc(&pose_, 96);
    }
    ~ResidentFrameTracker() {
        for (void* p : {res_, meas_, src_, flags_, count_, pose_})
            if (p) ptam_dev_free(c_.handle(), p);
    }
    ResidentFrameTracker(const ResidentFrameTracker&) = delete;
    ResidentFrameTracker& operator=(const ResidentFrameTracker&) = delete;

    // d_world: 3 doubles per query every nWorldStrideBytes (sizeof(ptam_pvs_point) reads them out of the PVS input)
    SE3 SearchAndUpdatePose(KeyFrame& kfCurrent, int n, const ptam_patch_query* d_queries, const uint8_t* d_templates,
                            const void* d_world, int nWorldStrideBytes, const SE3& se3Predicted, int nEstimator = PTAM_EST_TUKEY) {
        check(ptam_find_patch_coarse_batch_dev(c_.handle(), kfCurrent.handle(), n, d_queries, d_templates,
                                               (ptam_patch_result*)res_), "ptam_find_patch_coarse_batch_dev");
        int32_t* cnt = (int32_t*)count_;
        check(ptam_gather_pose_meas_dev(c_.handle(), n, d_queries, (const ptam_patch_result*)res_, nullptr, d_world,
                                        nWorldStrideBytes, (ptam_pose_meas*)meas_, (int32_t*)src_, cnt, cnt + 2),
              "ptam_gather_pose_meas_dev");
        ptam_gn_opts o;
        ptam_gn_opts_default(&o);
        o.estimator = nEstimator;
        double in[12], out[12];
        se3Predicted.to12(in);
        check(ptam_pose_gn_dev_counted(c_.handle(), n, cnt, (const ptam_pose_meas*)meas_, nullptr, (double*)pose_, &o,
                                       (int32_t*)flags_, nullptr, in, out), "ptam_pose_gn_dev_counted");
        return SE3::from12(out);
    }
    // measurements of the last frame: count, manMeasFound per level (src/Tracker.cc:892), and per measurement the query it
    // came from and whether iteration 9 weighted it to zero
    int ReadBack(std::vector<int32_t>& vnSourceQuery, std::vector<int32_t>& vnOutlier, int anMeasFound[PTAM_LEVELS] = nullptr) {
        int32_t head[8];
        check(ptam_dev_download(c_.handle(), head, count_, sizeof head), "ptam_dev_download");
        const int n = head[0];
        vnSourceQuery.resize(n);
        vnOutlier.resize(n);
        if (n > 0) {
            check(ptam_dev_download(c_.handle(), vnSourceQuery.data(), src_, 4 * (size_t)n), "ptam_dev_download");
            check(ptam_dev_download(c_.handle(), vnOutlier.data(), flags_, 4 * (size_t)n), "ptam_dev_download");
        }
        if (anMeasFound)
            for (int l = 0; l < PTAM_LEVELS; l++) anMeasFound[l] = head[2 + l];
        return n;
    }

private:
    void alloc(void** p, size_t bytes) { check(ptam_dev_alloc(c_.handle(), bytes, p), "ptam_dev_alloc"); }
    Context& c_;
    int cap_;
    void *res_ = nullptr, *meas_ = nullptr, *src_ = nullptr, *flags_ = nullptr, *count_ = nullptr, *pose_ = nullptr;
};

// Tracker::TrackMap (src/Tracker.cc:442-696) as one device-resident chain: the map (world positions, pixel vectors, patch
// sources) lives on the device between frames; a frame is one call that returns the refined pose, mbDidCoarse, the
// per-level manMeasAttempted / manMeasFound counters and the scene-depth sums.  The caller keeps what is host logic in the
// reference: the bTryCoarse heuristics (:505-516), the motion model, the tracking-quality assessment, and the two random
// orders per frame (SetShuffle: permutations of the map indices that replace std::random_shuffle, :483-484 and :598).
class MapTracker {
public:
    MapTracker(Context& c, int nMaxPoints) : c_(c) { check(ptam_tracker_create(c.handle(), nMaxPoints, &h_), "ptam_tracker_create"); }
    ~MapTracker() { ptam_tracker_destroy(h_); }
    MapTracker(const MapTracker&) = delete;
    MapTracker& operator=(const MapTracker&) = delete;
    // vSources[i]: src_kf / src_level / center_x / center_y of map point i (MapPoint::pPatchSourceKF, nSourceLevel, irCenter)
    void SetMap(const std::vector<ptam_pvs_point>& vMapPoints, const std::vector<ptam_template_query>& vSources) {
        check(ptam_tracker_set_map(h_, (int)vMapPoints.size(), vMapPoints.data(), vSources.data()), "ptam_tracker_set_map");
    }
    // The map after the mapmaker changed it: vPrevIndex[i] = index of point i in the map handed over last time, -1 for a new
    // point.  Persisting points keep their TrackerData (PatchFinder template, warp, mbTemplateBad) as in the reference.
    void UpdateMap(const std::vector<ptam_pvs_point>& vMapPoints, const std::vector<ptam_template_query>& vSources,
                   const std::vector<int32_t>& vPrevIndex) {
        if (vSources.size() != vMapPoints.size() || vPrevIndex.size() != vMapPoints.size()) throw std::runtime_error("UpdateMap: sizes differ");
        check(ptam_tracker_update_map(h_, (int)vMapPoints.size(), vMapPoints.data(), vSources.data(), vPrevIndex.data()), "ptam_tracker_update_map");
    }
    void SetShuffle(const std::vector<int32_t>& vLevels, const std::vector<int32_t>& vFine) {
        check(ptam_tracker_set_shuffle(h_, vLevels.data(), vFine.data()), "ptam_tracker_set_shuffle");
    }
    ptam_trackmap_result TrackMap(KeyFrame& kfCurrent, const SE3& se3Predicted, const ptam_trackmap_opts* pOpts = nullptr) {
        double in[12];
        se3Predicted.to12(in);
        ptam_trackmap_result r;
        check(ptam_track_map(h_, kfCurrent.handle(), in, pOpts, &r), "ptam_track_map");
        return r;
    }
    // The frame arrives with its image (device-resident, stride == width): MakeKeyFrame_Lite of it into kfCurrent + TrackMap
    ptam_trackmap_result TrackFrame(KeyFrame& kfCurrent, const uint8_t* dFrame, const SE3& se3Predicted, const ptam_trackmap_opts* pOpts = nullptr) {
        double in[12];
        se3Predicted.to12(in);
        ptam_trackmap_result r;
        check(ptam_track_map_frame(h_, kfCurrent.handle(), dFrame, in, pOpts, &r), "ptam_track_map_frame");
        return r;
    }
    // The tracking branch of Tracker::TrackFrame (src/Tracker.cc:94, :134-137) for a camera that moves: MakeKeyFrame_Lite of the
    // device-resident frame, PredictPoseWithMotionModel, the bTryCoarse heuristics, TrackMap, UpdateMotionModel.  `motion` is the
    // tracker's model (ptam_motion_reset at Tracker::Reset; motion.just_recovered = 1 after a recovery); motion.pose is
    // mse3CamFromWorld afterwards.
    ptam_trackmap_result TrackFrame(KeyFrame& kfCurrent, const uint8_t* dFrame, ptam_motion_model& motion, const ptam_trackmap_opts* pOpts = nullptr) {
        ptam_trackmap_result r;
        check(ptam_track_frame(h_, kfCurrent.handle(), dFrame, &motion, pOpts, &r), "ptam_track_frame");
        return r;
    }
    // Several cameras (or agents) on one device: ONE chain of launches for all their frames, results as the single calls give
    // them (ptam_track_map_frames_batch).  The trackers live in different Contexts with the same camera model and image size;
    // SetShuffle each of them first.
    static std::vector<ptam_trackmap_result> TrackFramesBatch(const std::vector<MapTracker*>& vTrackers, const std::vector<KeyFrame*>& vCurrent,
                                                              const std::vector<const uint8_t*>& vdFrames, const std::vector<SE3>& vPredicted,
                                                              const ptam_trackmap_opts* pOpts = nullptr) {
        const size_t n = vTrackers.size();
        if (n == 0 || vCurrent.size() != n || vdFrames.size() != n || vPredicted.size() != n) throw std::runtime_error("TrackFramesBatch: sizes differ");
        std::vector<ptam_tracker*> t(n);
        std::vector<ptam_kf*> k(n);
        std::vector<double> poses(12 * n);
        for (size_t i = 0; i < n; i++) {
            t[i] = vTrackers[i]->h_;
            k[i] = vCurrent[i]->handle();
            vPredicted[i].to12(&poses[12 * i]);
        }
        std::vector<ptam_trackmap_result> r(n);
        check(ptam_track_map_frames_batch((int)n, t.data(), k.data(), vdFrames.data(), poses.data(), pOpts, r.data()), "ptam_track_map_frames_batch");
        return r;
    }
    // vIterationSet of the last frame: what :667-676 turns into mCurrentKF.mMeasurements and what routes the outlier flags
    // back to MapPoint::nMEstimatorOutlierCount
    std::vector<ptam_trackmap_meas> IterationSet() {
        int n = 0;
        check(ptam_tracker_read_iteration_set(h_, nullptr, 0, &n), "ptam_tracker_read_iteration_set");
        std::vector<ptam_trackmap_meas> v((size_t)n);
        if (n > 0) check(ptam_tracker_read_iteration_set(h_, v.data(), n, &n), "ptam_tracker_read_iteration_set");
        return v;
    }

private:
    Context& c_;
    ptam_tracker* h_ = nullptr;
};

// MapMaker::ReFind_Common (src/MapMaker.cc:943-1020) over the candidate points of one keyframe (the loop body of
// ReFindInSingleKeyFrame :1027-1042): out[i].found -> add Measurement{nLevel = level, v2RootPos = root_pos, bSubPix =
// sub_pix, Source = SRC_REFIND} and insert into sMeasurementKFs; out[i].never_retry -> insert into sNeverRetryKFs.
// The caller filters the points exactly as :947-948 does (already measured / never retry) before the call.
inline std::vector<ptam_refind_result> ReFindInKeyFrame(Context& c, KeyFrame& k, const SE3& se3CfromW,
                                                        const std::vector<ptam_pvs_point>& vPoints,
                                                        const std::vector<ptam_template_query>& vSources) {
    double pose[12];
    se3CfromW.to12(pose);
    std::vector<ptam_refind_result> out(vPoints.size());
    check(ptam_refind_batch(c.handle(), k.handle(), pose, (int)vPoints.size(), vPoints.data(), vSources.data(), out.data()),
          "ptam_refind_batch");
    return out;
}

// The `static PatchFinder Finder` of MapMaker::ReFind_Common (src/MapMaker.cc:977) and the loops that drive it pair by pair:
// ReFindNewlyMade (:1046-1066: one new point against every keyframe in turn) and ReFindFromFailureQueue (:1070-1082: the
// sorted (keyframe, point) pairs).  The caller lists the pairs in the reference's order — with `skip` set where :947-948
// would return at once — and applies out[i] as for ReFindInKeyFrame above.  One ReFinder per MapMaker: its state (last
// template, last warp, mbTemplateBad) carries from call to call exactly like the static's.
class ReFinder {
public:
    explicit ReFinder(Context& c) : c_(c) { check(ptam_refinder_create(c.handle(), &h_), "ptam_refinder_create"); }
    ~ReFinder() { ptam_refinder_destroy(h_); }
    ReFinder(const ReFinder&) = delete;
    ReFinder& operator=(const ReFinder&) = delete;
    std::vector<ptam_refind_result> Find(const std::vector<ptam_refind_pair>& vPairs, std::vector<int32_t>* pvTemplateKept = nullptr) {
        std::vector<ptam_refind_result> out(vPairs.size());
        if (pvTemplateKept) pvTemplateKept->assign(vPairs.size(), 0);
        check(ptam_refind_pairs(c_.handle(), h_, (int)vPairs.size(), vPairs.data(), out.data(), pvTemplateKept ? pvTemplateKept->data() : nullptr),
              "ptam_refind_pairs");
        return out;
    }

private:
    Context& c_;
    ptam_refinder* h_ = nullptr;
};

// class Bundle (include/Bundle.h:106-152)
class Bundle {
public:
    explicit Bundle(Context& c, const ptam_ba_opts* opts = nullptr) { check(ptam_ba_create(c.handle(), opts, &h_), "ptam_ba_create"); }
    ~Bundle() { ptam_ba_destroy(h_); }
    Bundle(const Bundle&) = delete;
    Bundle& operator=(const Bundle&) = delete;
    int AddCamera(const SE3& se3CamFromWorld, bool bFixed) {   // :111
        double p[12];
        se3CamFromWorld.to12(p);
        int id = ptam_ba_add_camera(h_, p, bFixed);
        check(id, "ptam_ba_add_camera");
        return id;
    }
    int AddPoint(const Vec<3>& v3Pos) {   // :112
        int id = ptam_ba_add_point(h_, v3Pos.data());
        check(id, "ptam_ba_add_point");
        return id;
    }
    void AddMeas(int nCam, int nPoint, const Vec<2>& v2Pos, double dSigmaSquared) {   // :113
        check(ptam_ba_add_meas(h_, nCam, nPoint, v2Pos.data(), dSigmaSquared), "ptam_ba_add_meas");
    }
    // :114 ; returns mnAccepted (>= 0).  The reference's own -1 is dead code (Do_LM_Step returns true unconditionally,
    // src/Bundle.cc:550), so a negative value only leaves this shim for the one input the device path refuses: a duplicated
    // (camera, point) measurement (PTAM_E_ARG; the reference silently keeps the last one).  MapMaker treats a
    // negative return as "ditch the map" (src/MapMaker.cc:887-892) — harsh, but it keeps its thread alive, where an
    // exception thrown out of Compute() would not.  HIP / communicator failures still throw.
    int Compute(bool* pbAbortSignal) {
        int acc = 0;
        const int rc = ptam_ba_compute(h_, reinterpret_cast<const volatile unsigned char*>(pbAbortSignal), &acc);
        if (rc == PTAM_E_ARG) return -1;
        check(rc, "ptam_ba_compute");
        return acc;
    }
    bool Converged() const { return ptam_ba_converged(h_) != 0; }   // :115
    Vec<3> GetPoint(int n) const {                                   // :116
        Vec<3> v{};
        check(ptam_ba_get_point(h_, n, v.data()), "ptam_ba_get_point");
        return v;
    }
    SE3 GetCamera(int n) const {   // :117
        double p[12];
        check(ptam_ba_get_camera(h_, n, p), "ptam_ba_get_camera");
        return SE3::from12(p);
    }
    std::vector<std::pair<int, int>> GetOutlierMeasurements() const {   // :118 ; (point, camera)
        const int n = ptam_ba_get_outliers(h_, nullptr, 0);
        std::vector<int32_t> raw((size_t)2 * (n > 0 ? n : 0));
        if (n > 0) ptam_ba_get_outliers(h_, raw.data(), n);
        std::vector<std::pair<int, int>> out;
        for (int i = 0; i < n; i++) out.emplace_back(raw[2 * i], raw[2 * i + 1]);
        return out;
    }
    ptam_ba* handle() const { return h_; }

private:
    ptam_ba* h_ = nullptr;
};

}   // namespace ptam
#endif
