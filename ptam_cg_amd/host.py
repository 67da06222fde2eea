"""Host-side mirror of the reference's class surface for the hot path, over the C ABI.

Class / method names follow the reference (KeyFrame::MakeKeyFrame_Lite src/KeyFrame.cc:18,
PatchFinder::FindPatchCoarse src/PatchFinder.cc:160, Tracker::CalcPoseUpdate src/Tracker.cc:928,
Bundle include/Bundle.h:106-152) so that tests read like tests of the reference.  Every class takes
the bound library (`_abi.Bound`) it drives; `ptam_cg_amd.Context()` defaults to libptam_hip.so.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import (BaOpts, BaTrial, CamParams, GnOpts, Int2, MotionModel, PatchQuery, PatchResult, PoseMeas,
                   PoseUpdateMeas, Projection)

# config/camera.cfg:7
DEFAULT_CAMERA = (1.0803, 1.43987, 0.519983, 0.548655, 0.244943)

PATCH_QUERY_DT = np.dtype([("x", "<i4"), ("y", "<i4"), ("level", "<i4"), ("range", "<u4")])
PATCH_RESULT_DT = np.dtype([("found", "<i4"), ("best_ssd", "<i4"), ("best_x", "<i4"), ("best_y", "<i4"),
                            ("n_scored", "<i4"), ("pad_", "<i4"), ("pos", "<f8", (2,))])
PROJECTION_DT = np.dtype([("cam", "<f8", (3,)), ("image", "<f8", (2,)), ("derivs", "<f8", (4,)),
                          ("in_image", "<i4"), ("pad_", "<i4")])
POSE_MEAS_DT = np.dtype([("world", "<f8", (3,)), ("found", "<f8", (2,)), ("sqrt_inv_noise", "<f8")])
POSE_UPDATE_MEAS_DT = np.dtype([("found", "<f8", (2,)), ("image", "<f8", (2,)), ("sqrt_inv_noise", "<f8"),
                                ("jac", "<f8", (12,))])
SUBPIX_QUERY_DT = np.dtype([("coarse_pos", "<f8", (2,)), ("level", "<i4"), ("max_its", "<i4")])
SUBPIX_RESULT_DT = np.dtype([("converged", "<i4"), ("iterations", "<i4"), ("pos", "<f8", (2,)), ("mean_diff", "<f8")])
TEMPLATE_QUERY_DT = np.dtype([("src_kf", "<u8"), ("src_level", "<i4"), ("search_level", "<i4"), ("center_x", "<i4"),
                              ("center_y", "<i4"), ("warp_inverse", "<f8", (4,))])
EPIPOLAR_QUERY_DT = np.dtype([("level_x", "<i4"), ("level_y", "<i4"), ("normal", "<f8", (2,)), ("norm_dist", "<f8"),
                              ("along", "<f8", (2,)), ("min_len", "<f8"), ("max_len", "<f8"), ("max_dist_sq", "<f8")])
EPIPOLAR_RESULT_DT = np.dtype([("best", "<i4"), ("best_zmssd", "<i4"), ("n_scored", "<i4"), ("template_bad", "<i4")])
TEMPLATE_RESULT_DT = np.dtype([("bad", "<i4"), ("n_outside", "<i4"), ("sum", "<i4"), ("sum_sq", "<i4"), ("m2", "<f8", (4,))])
PVS_POINT_DT = np.dtype([("world", "<f8", (3,)), ("pixel_right_w", "<f8", (3,)), ("pixel_down_w", "<f8", (3,))])
PVS_RESULT_DT = np.dtype([("proj", PROJECTION_DT), ("warp_inverse", "<f8", (4,)), ("level", "<i4"), ("pad_", "<i4")])
BA_TRIAL_DT = np.dtype([("lambda", "<f8"), ("sigma_sq", "<f8"), ("err_old", "<f8"), ("err_new", "<f8"),
                        ("sum_sq_update", "<f8"), ("n_bad", "<i4"), ("accepted", "<i4")])
REFIND_RESULT_DT = np.dtype([("found", "<i4"), ("level", "<i4"), ("sub_pix", "<i4"), ("never_retry", "<i4"), ("root_pos", "<f8", (2,))])
REFIND_PAIR_DT = np.dtype([("kf", "<u8"), ("kf_pose", "<f8", (12,)), ("point", PVS_POINT_DT), ("source", TEMPLATE_QUERY_DT),
                           ("point_id", "<i8"), ("skip", "<i4"), ("pad_", "<i4")])
assert REFIND_PAIR_DT.itemsize == 248
TRACKMAP_OPTS_DT = np.dtype([("try_coarse", "<i4"), ("coarse_min", "<u4"), ("coarse_max", "<u4"), ("coarse_range", "<u4"),
                             ("coarse_subpix_its", "<i4"), ("max_patches", "<i4"), ("estimator", "<i4"), ("pad_", "<i4")])
TRACKMAP_RESULT_DT = np.dtype([("pose", "<f8", (12,)), ("did_coarse", "<i4"), ("n_pvs", "<i4", (4,)), ("attempted", "<i4", (4,)),
                               ("found", "<i4", (4,)), ("n_coarse", "<i4"), ("n_top", "<i4"), ("n_fine", "<i4"), ("n_meas", "<i4"),
                               ("depth_n", "<i4"), ("templates_reused", "<i4"), ("pad_", "<i4"), ("depth_sum", "<f8"), ("depth_sum_sq", "<f8")])
TRACKMAP_MEAS_DT = np.dtype([("point", "<i4"), ("level", "<i4"), ("found", "<i4"), ("did_subpix", "<i4"), ("outlier", "<i4"),
                             ("pad_", "<i4"), ("v2_found", "<f8", (2,))])
assert TRACKMAP_RESULT_DT.itemsize == 96 + 4 * 20 + 16 and TRACKMAP_MEAS_DT.itemsize == 40
assert PATCH_RESULT_DT.itemsize == C.sizeof(PatchResult)
assert PROJECTION_DT.itemsize == C.sizeof(Projection)
assert POSE_MEAS_DT.itemsize == C.sizeof(PoseMeas)
assert POSE_UPDATE_MEAS_DT.itemsize == C.sizeof(PoseUpdateMeas)
assert BA_TRIAL_DT.itemsize == C.sizeof(BaTrial)


class PtamError(RuntimeError):
    pass


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    """One per calling thread: camera model (ATANCamera, src/ATANCamera.cc) + stream + scratch."""

    def __init__(self, lib=None, camera=DEFAULT_CAMERA, size=(640, 480), device=0,
                 halfsample=_abi.HALFSAMPLE_R):
        if lib is None:
            from ._lib import load
            lib = load()
        self.lib = lib
        self.device = device
        self.size = tuple(size)
        self.cam = CamParams(*camera, size[0], size[1])
        h = C.c_void_p()
        self._check(lib.ctx_create(C.byref(self.cam), device, C.byref(h)), "ctx_create")
        self.h = h
        self._check(lib.ctx_set_halfsample(self.h, halfsample), "ctx_set_halfsample")

    def _check(self, rc, what):
        if rc < 0:
            msg = self.lib.last_error() if self.lib.has("last_error") else b""
            raise PtamError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        return rc

    def sync(self):
        self._check(self.lib.ctx_sync(self.h), "ctx_sync")

    def cache_hazards(self):
        """pose-loop re-projections that bailed out before the camera model (ptam_ctx_cache_hazards: where the reference's
        ProjectAndDerivs would read another point's cached derivatives)"""
        v = C.c_longlong()
        self._check(self.lib.ctx_cache_hazards(self.h, C.byref(v)), "ctx_cache_hazards")
        return v.value

    def camera_constants(self):
        out = np.zeros(8)
        self._check(self.lib.ctx_camera_constants(self.h, _pd(out)), "camera_constants")
        return dict(zip(["focal_x", "focal_y", "centre_x", "centre_y", "two_tan", "w_inv",
                         "largest_radius", "max_r"], out))

    def one_pixel_dist(self):
        """ATANCamera::OnePixelDist() (src/ATANCamera.cc:69-75)"""
        out = np.zeros(1)
        self._check(self.lib.ctx_one_pixel_dist(self.h, _pd(out)), "one_pixel_dist")
        return float(out[0])

    def close(self):
        if getattr(self, "h", None):
            self.lib.ctx_destroy(self.h)
            self.h = None

    # -- TrackerData::Project batch (include/Tracker.h:70-94) --
    def project_points(self, world, pose):
        world = np.ascontiguousarray(world, dtype=np.float64).reshape(-1, 3)
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        out = np.zeros(len(world), dtype=PROJECTION_DT)
        self._check(self.lib.project_points(self.h, len(world), _ptr(world), _pd(pose), _ptr(out)),
                    "project_points")
        return out

    # -- TrackMap PVS loop + CalcSearchLevelAndWarpMatrix (src/Tracker.cc:453-478, src/PatchFinder.cc:52-84) --
    def track_pvs(self, world, pixel_right_w, pixel_down_w, pose):
        n = len(world)
        pts = np.zeros(n, dtype=PVS_POINT_DT)
        pts["world"], pts["pixel_right_w"], pts["pixel_down_w"] = world, pixel_right_w, pixel_down_w
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        out = np.zeros(n, dtype=PVS_RESULT_DT)
        counts = np.zeros(4, dtype=np.int32)
        self._check(self.lib.track_pvs(self.h, n, _ptr(pts), _pd(pose), _ptr(out), _ptr(counts)), "track_pvs")
        return out, counts

    # -- Tracker pose Gauss-Newton (src/Tracker.cc:613-643) --
    def gn_opts(self, **kw):
        o = GnOpts()
        self.lib.gn_opts_default(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def pose_gn(self, world, found, sqrt_inv_noise, pose, opts=None, entry=None):
        n = len(world)
        meas = np.zeros(n, dtype=POSE_MEAS_DT)
        meas["world"] = world
        meas["found"] = found
        meas["sqrt_inv_noise"] = sqrt_inv_noise
        pose = np.array(pose, dtype=np.float64).reshape(12).copy()
        opts = opts or self.gn_opts()
        flags = np.zeros(n, dtype=np.int32)
        updates = np.zeros((opts.iterations, 6))
        if entry is not None:
            entry = np.ascontiguousarray(entry, dtype=PROJECTION_DT)
        self._check(self.lib.pose_gn(self.h, n, _ptr(meas), _ptr(entry), _pd(pose), C.byref(opts),
                                     _ptr(flags), _ptr(updates)), "pose_gn")
        return pose, flags, updates

    def pose_gn_state(self, world, found, sqrt_inv_noise, pose, opts=None, entry=None):
        """pose_gn that also returns the measurements' TrackerData state at loop exit (ptam_pose_gn_state)"""
        n = len(world)
        meas = np.zeros(n, dtype=POSE_MEAS_DT)
        meas["world"], meas["found"], meas["sqrt_inv_noise"] = world, found, sqrt_inv_noise
        pose = np.array(pose, dtype=np.float64).reshape(12).copy()
        opts = opts or self.gn_opts()
        flags = np.zeros(n, dtype=np.int32)
        updates = np.zeros((opts.iterations, 6))
        state = np.zeros(n, dtype=PROJECTION_DT)
        if entry is not None:
            entry = np.ascontiguousarray(entry, dtype=PROJECTION_DT)
        self._check(self.lib.pose_gn_state(self.h, n, _ptr(meas), _ptr(entry), _pd(pose), C.byref(opts), _ptr(flags),
                                           _ptr(updates), _ptr(state)), "pose_gn_state")
        return pose, flags, updates, state

    def reproject_points(self, world, pose, state):
        """TrackerData::Project on existing state (ptam_reproject_points) -> updated copy of `state`"""
        world = np.ascontiguousarray(world, dtype=np.float64).reshape(-1, 3)
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        st = np.ascontiguousarray(state, dtype=PROJECTION_DT).copy()
        self._check(self.lib.reproject_points(self.h, len(world), _ptr(world), _pd(pose), _ptr(st)), "reproject_points")
        return st

    # -- Tracker::CalcPoseUpdate (src/Tracker.cc:928-1005) --
    def calc_pose_update(self, found, image, sqrt_inv_noise, jac, override_sigma_sq=0.0,
                         estimator=_abi.EST_TUKEY, prior=100.0):
        n = len(found)
        meas = np.zeros(n, dtype=POSE_UPDATE_MEAS_DT)
        meas["found"], meas["image"] = found, image
        meas["sqrt_inv_noise"] = sqrt_inv_noise
        meas["jac"] = np.asarray(jac).reshape(n, 12)
        mu = np.zeros(6)
        flags = np.zeros(n, dtype=np.int32)
        self._check(self.lib.calc_pose_update(self.h, n, _ptr(meas), override_sigma_sq, estimator, prior,
                                              _pd(mu), _ptr(flags)), "calc_pose_update")
        return mu, flags


class KeyFrame:
    """KeyFrame (include/KeyFrame.h:130-149): 4 pyramid levels with FAST corners + row LUTs."""

    def __init__(self, ctx, handle=None):
        self.ctx, self.lib = ctx, ctx.lib
        w, h = ctx.size
        if handle is None:
            handle = C.c_void_p()
            ctx._check(self.lib.kf_create(ctx.h, w, h, C.byref(handle)), "kf_create")
        self.h = handle

    def MakeKeyFrame_Lite(self, im):
        im = np.ascontiguousarray(im, dtype=np.uint8)
        assert im.shape == (self.ctx.size[1], self.ctx.size[0]), im.shape
        self.ctx._check(self.lib.make_keyframe_lite(self.ctx.h, self.h, _ptr(im), im.strides[0]),
                        "make_keyframe_lite")
        return self

    def MakeKeyFrame_Rest(self, min_shi_tomasi=70.0):
        """fast_nonmax + Shi-Tomasi candidates (src/KeyFrame.cc:61-82); -> list of 4 dicts
        {max_corners (n,2), st_scores (n,), candidates (m,2), candidate_scores (m,)}"""
        self.ctx._check(self.lib.make_keyframe_rest(self.ctx.h, self.h), "make_keyframe_rest")
        out = []
        for l in range(_abi.LEVELS):
            n = C.c_int()
            self.ctx._check(self.lib.kf_rest_info(self.ctx.h, self.h, l, C.byref(n)), "kf_rest_info")
            mc = np.zeros((n.value, 2), dtype=np.int32)
            st = np.zeros(n.value, dtype=np.float64)
            self.ctx._check(self.lib.kf_read_rest(self.ctx.h, self.h, l, _ptr(mc), _ptr(st)), "kf_read_rest")
            sel = st > min_shi_tomasi
            out.append({"max_corners": mc, "st_scores": st, "candidates": mc[sel], "candidate_scores": st[sel]})
        return out

    def clone(self):
        out = C.c_void_p()
        self.ctx._check(self.lib.kf_clone(self.ctx.h, self.h, C.byref(out)), "kf_clone")
        return KeyFrame(self.ctx, out)

    def level(self, l):
        """-> dict(im, corners (n,2) int32 [x,y] raster order, rowlut)"""
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        self.ctx._check(self.lib.kf_level_info(self.ctx.h, self.h, l, C.byref(w), C.byref(h), C.byref(n)),
                        "kf_level_info")
        im = np.zeros((h.value, w.value), dtype=np.uint8)
        corners = np.zeros((n.value, 2), dtype=np.int32)
        lut = np.zeros(h.value, dtype=np.int32)
        self.ctx._check(self.lib.kf_read_level(self.ctx.h, self.h, l, _ptr(im), _ptr(corners), _ptr(lut)),
                        "kf_read_level")
        return {"im": im, "corners": corners, "rowlut": lut}

    def implane_corners(self, l):
        """Level::vImplaneCorners (src/MapMaker.cc:605-614): (n, 2) float64"""
        n = C.c_int()
        self.ctx._check(self.lib.kf_implane_corners(self.ctx.h, self.h, l, None, 0, C.byref(n)), "kf_implane_corners")
        out = np.zeros((n.value, 2))
        self.ctx._check(self.lib.kf_implane_corners(self.ctx.h, self.h, l, _ptr(out), n.value, None), "kf_implane_corners")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.kf_destroy(self.h)
            self.h = None


class PatchFinder:
    """Batched PatchFinder::FindPatchCoarse (src/PatchFinder.cc:160-211) over one KeyFrame."""

    def __init__(self, ctx):
        self.ctx, self.lib = ctx, ctx.lib

    def FindPatchCoarse(self, kf, queries, templates):
        queries = np.ascontiguousarray(queries, dtype=PATCH_QUERY_DT)
        templates = np.ascontiguousarray(templates, dtype=np.uint8).reshape(len(queries), 64)
        res = np.zeros(len(queries), dtype=PATCH_RESULT_DT)
        self.ctx._check(self.lib.find_patch_coarse_batch(self.ctx.h, kf.h, len(queries), _ptr(queries),
                                                         _ptr(templates), _ptr(res)), "find_patch_coarse")
        return res

    def SubPix(self, kf, coarse_pos, levels, templates, max_its=8):
        """MakeSubPixTemplate + IterateSubPixToConvergence for a batch (src/PatchFinder.cc:219-318)"""
        n = len(coarse_pos)
        q = np.zeros(n, dtype=SUBPIX_QUERY_DT)
        q["coarse_pos"], q["level"], q["max_its"] = coarse_pos, levels, max_its
        templates = np.ascontiguousarray(templates, dtype=np.uint8).reshape(n, 64)
        res = np.zeros(n, dtype=SUBPIX_RESULT_DT)
        self.ctx._check(self.lib.subpix_batch(self.ctx.h, kf.h, n, _ptr(q), _ptr(templates), _ptr(res)), "subpix_batch")
        return res

    def MakeTemplateCoarseCont(self, src_kfs, src_levels, centers, search_levels, warp_inverses):
        """PatchFinder::MakeTemplateCoarseCont for a batch (src/PatchFinder.cc:98-127): `src_kfs` is one KeyFrame or a
        sequence of them (MapPoint::pPatchSourceKF).  Returns (templates uint8 [n,64], results).  The reference's
        "same point, warp moved < 0.07" reuse test is the caller's (it needs the previous call's state)."""
        n = len(src_levels)
        q = np.zeros(n, dtype=TEMPLATE_QUERY_DT)
        kfs = [src_kfs] * n if isinstance(src_kfs, KeyFrame) else list(src_kfs)
        q["src_kf"] = [k.h.value if hasattr(k.h, "value") else int(k.h) for k in kfs]
        q["src_level"], q["search_level"] = src_levels, search_levels
        c = np.asarray(centers, dtype=np.int32).reshape(n, 2)
        q["center_x"], q["center_y"] = c[:, 0], c[:, 1]
        q["warp_inverse"] = np.asarray(warp_inverses, dtype=np.float64).reshape(n, 4)
        tm = np.zeros((n, 64), dtype=np.uint8)
        res = np.zeros(n, dtype=TEMPLATE_RESULT_DT)
        self.ctx._check(self.lib.make_templates_batch(self.ctx.h, n, _ptr(q), _ptr(tm), _ptr(res)), "make_templates_batch")
        return tm, res

    def ReFind(self, kf, kf_pose, world, pixel_right_w, pixel_down_w, src_kfs, src_levels, centers):
        """MapMaker::ReFind_Common for a batch of map points against one keyframe (src/MapMaker.cc:943-1020)"""
        n = len(world)
        pts = np.zeros(n, dtype=PVS_POINT_DT)
        pts["world"], pts["pixel_right_w"], pts["pixel_down_w"] = world, pixel_right_w, pixel_down_w
        q = np.zeros(n, dtype=TEMPLATE_QUERY_DT)
        kfs = [src_kfs] * n if isinstance(src_kfs, KeyFrame) else list(src_kfs)
        q["src_kf"] = [k.h.value if hasattr(k.h, "value") else int(k.h) for k in kfs]
        q["src_level"] = src_levels
        c = np.asarray(centers, dtype=np.int32).reshape(n, 2)
        q["center_x"], q["center_y"] = c[:, 0], c[:, 1]
        pose = np.ascontiguousarray(kf_pose, dtype=np.float64).reshape(12)
        out = np.zeros(n, dtype=REFIND_RESULT_DT)
        self.ctx._check(self.lib.refind_batch(self.ctx.h, kf.h, _pd(pose), n, _ptr(pts), _ptr(q), _ptr(out)), "refind_batch")
        return out

    def EpipolarSearch(self, src_kf, target_kf, level, queries):
        """the corner scan of MapMaker::AddPointEpipolar (src/MapMaker.cc:598-637) for a batch of candidates"""
        q = np.ascontiguousarray(queries, dtype=EPIPOLAR_QUERY_DT)
        res = np.zeros(len(q), dtype=EPIPOLAR_RESULT_DT)
        self.ctx._check(self.lib.epipolar_search_batch(self.ctx.h, src_kf.h, target_kf.h, level, len(q), _ptr(q), _ptr(res)),
                        "epipolar_search_batch")
        return res

    def ZMSSDAtPoint(self, kf, level, points, template):
        points = np.ascontiguousarray(points, dtype=np.int32).reshape(-1, 2)
        template = np.ascontiguousarray(template, dtype=np.uint8).reshape(64)
        out = np.zeros(len(points), dtype=np.int32)
        self.ctx._check(self.lib.zmssd_at_points(self.ctx.h, kf.h, level, len(points), _ptr(points),
                                                 _ptr(template), _ptr(out)), "zmssd_at_points")
        return out


class DevBuf:
    """A device allocation of the context (ptam_dev_alloc / upload / download / free)."""

    def __init__(self, ctx, nbytes_or_array):
        self.ctx = ctx
        arr = nbytes_or_array if isinstance(nbytes_or_array, np.ndarray) else None
        self.nbytes = int(arr.nbytes if arr is not None else nbytes_or_array)
        self.p = C.c_void_p()
        ctx._check(ctx.lib.dev_alloc(ctx.h, max(self.nbytes, 8), C.byref(self.p)), "dev_alloc")
        if arr is not None and arr.nbytes:
            self.upload(arr)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        self.ctx._check(self.ctx.lib.dev_upload(self.ctx.h, self.p, arr.ctypes.data, arr.nbytes), "dev_upload")

    def download(self, dtype, count):
        out = np.zeros(count, dtype=dtype)
        if out.nbytes:
            self.ctx._check(self.ctx.lib.dev_download(self.ctx.h, out.ctypes.data, self.p, out.nbytes), "dev_download")
        return out

    def free(self):
        if self.p:
            self.ctx.lib.dev_free(self.ctx.h, self.p)
            self.p = None


class ReFinder:
    """MapMaker::ReFind_Common through ONE PatchFinder (`static PatchFinder Finder`, src/MapMaker.cc:977), pair after pair in
    the caller's order (ReFindNewlyMade :1046-1066, ReFindFromFailureQueue :1070-1082): ptam_refinder_* / ptam_refind_pairs"""

    def __init__(self, ctx):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        ctx._check(self.lib.refinder_create(ctx.h, C.byref(self.h)), "refinder_create")

    @staticmethod
    def pairs(kfs, kf_poses, world, pixel_right_w, pixel_down_w, src_kfs, src_levels, centers, point_ids, skip=None):
        """one record per (keyframe, point) pair, all arguments per pair"""
        n = len(kfs)
        raw = lambda k: k.h.value if hasattr(k.h, "value") else int(k.h)
        p = np.zeros(n, dtype=REFIND_PAIR_DT)
        p["kf"] = [raw(k) for k in kfs]
        p["kf_pose"] = np.asarray(kf_poses, dtype=np.float64).reshape(n, 12)
        p["point"]["world"], p["point"]["pixel_right_w"], p["point"]["pixel_down_w"] = world, pixel_right_w, pixel_down_w
        src = [src_kfs] * n if isinstance(src_kfs, KeyFrame) else list(src_kfs)
        p["source"]["src_kf"] = [raw(k) for k in src]
        p["source"]["src_level"] = src_levels
        c = np.asarray(centers, dtype=np.int32).reshape(n, 2)
        p["source"]["center_x"], p["source"]["center_y"] = c[:, 0], c[:, 1]
        p["point_id"] = point_ids
        if skip is not None:
            p["skip"] = skip
        return p

    def find(self, pairs):
        """-> (results REFIND_RESULT_DT[n], template_kept int32[n])"""
        pairs = np.ascontiguousarray(pairs, dtype=REFIND_PAIR_DT)
        out = np.zeros(len(pairs), dtype=REFIND_RESULT_DT)
        kept = np.zeros(len(pairs), dtype=np.int32)
        self.ctx._check(self.lib.refind_pairs(self.ctx.h, self.h, len(pairs), _ptr(pairs), _ptr(out), _ptr(kept)), "refind_pairs")
        return out, kept

    def close(self):
        if getattr(self, "h", None):
            self.lib.refinder_destroy(self.h)
            self.h = None


class Tracker:
    """Tracker::TrackMap (src/Tracker.cc:442-696) as one device-resident chain (ptam_tracker_* / ptam_track_map): the map
    stays on the device, a frame is one call."""

    def __init__(self, ctx, max_points):
        self.ctx, self.lib = ctx, ctx.lib
        self.h = C.c_void_p()
        ctx._check(self.lib.tracker_create(ctx.h, int(max_points), C.byref(self.h)), "tracker_create")
        self._keep = []

    def set_map(self, world, pixel_right_w, pixel_down_w, src_kfs, src_levels, centers, prev_index=None):
        """prev_index (update_map): for every point its index in the previous map, -1 for a new one — those points keep their
        PatchFinder state (ptam_tracker_update_map)"""
        n = len(world)
        pts = np.zeros(n, dtype=PVS_POINT_DT)
        pts["world"], pts["pixel_right_w"], pts["pixel_down_w"] = world, pixel_right_w, pixel_down_w
        q = np.zeros(n, dtype=TEMPLATE_QUERY_DT)
        kfs = [src_kfs] * n if isinstance(src_kfs, KeyFrame) else list(src_kfs)
        self._keep = kfs
        q["src_kf"] = [k.h.value if hasattr(k.h, "value") else int(k.h) for k in kfs]
        q["src_level"] = src_levels
        c = np.asarray(centers, dtype=np.int32).reshape(n, 2)
        q["center_x"], q["center_y"] = c[:, 0], c[:, 1]
        if prev_index is None:
            self.ctx._check(self.lib.tracker_set_map(self.h, n, _ptr(pts), _ptr(q)), "tracker_set_map")
        else:
            pi = np.ascontiguousarray(prev_index, dtype=np.int32)
            assert len(pi) == n
            self.ctx._check(self.lib.tracker_update_map(self.h, n, _ptr(pts), _ptr(q), _ptr(pi)), "tracker_update_map")
        self.n = n

    def update_map(self, world, pixel_right_w, pixel_down_w, src_kfs, src_levels, centers, prev_index):
        self.set_map(world, pixel_right_w, pixel_down_w, src_kfs, src_levels, centers, prev_index=prev_index)

    def set_shuffle(self, shuffle_levels, shuffle_fine):
        a = np.ascontiguousarray(shuffle_levels, dtype=np.int32)
        b = np.ascontiguousarray(shuffle_fine, dtype=np.int32)
        self.ctx._check(self.lib.tracker_set_shuffle(self.h, _ptr(a), _ptr(b)), "tracker_set_shuffle")

    def opts(self, **kw):
        o = np.zeros(1, dtype=TRACKMAP_OPTS_DT)
        self.lib.trackmap_opts_default(_ptr(o))
        for k, v in kw.items():
            o[k] = v
        return o

    def TrackMap(self, kf, pose, opts=None):
        """-> structured scalar (TRACKMAP_RESULT_DT)"""
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        res = np.zeros(1, dtype=TRACKMAP_RESULT_DT)
        self.ctx._check(self.lib.track_map(self.h, kf.h, _pd(pose), _ptr(opts) if opts is not None else None, _ptr(res)), "track_map")
        return res[0]

    def TrackFrame(self, kf, d_frame, pose, opts=None):
        """MakeKeyFrame_Lite of the device-resident frame (DevBuf or raw pointer) into `kf` + TrackMap in one call"""
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        res = np.zeros(1, dtype=TRACKMAP_RESULT_DT)
        dp = d_frame.p if isinstance(d_frame, DevBuf) else d_frame
        self.ctx._check(self.lib.track_map_frame(self.h, kf.h, dp, _pd(pose), _ptr(opts) if opts is not None else None, _ptr(res)),
                        "track_map_frame")
        return res[0]

    def motion_model(self, pose):
        """a ptam_motion_model at `pose` with zero velocity (Tracker::Reset, src/Tracker.cc:52-56)"""
        m = MotionModel()
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        self.lib.motion_reset(C.byref(m), _pd(pose))
        return m

    def TrackFrameMoving(self, kf, d_frame, motion, opts=None):
        """the tracking branch of Tracker::TrackFrame (src/Tracker.cc:94,134-137): keyframe of the frame, motion-model
        prediction, bTryCoarse heuristics, TrackMap, motion-model update — ptam_track_frame.  `motion` is updated in place."""
        res = np.zeros(1, dtype=TRACKMAP_RESULT_DT)
        dp = d_frame.p if isinstance(d_frame, DevBuf) else d_frame
        self.ctx._check(self.lib.track_frame(self.h, kf.h, dp, C.byref(motion), _ptr(opts) if opts is not None else None, _ptr(res)),
                        "track_frame")
        return res[0]

    def set_profiling(self, on=True):
        self.ctx._check(self.lib.tracker_set_profiling(self.h, 1 if on else 0), "tracker_set_profiling")

    def stage_times(self):
        """-> {stage: us per profiled frame} (ptam_tracker_stage_time)"""
        out = {}
        for i, name in enumerate(_abi.TRACK_STAGE_NAMES):
            ms, n = C.c_double(), C.c_int()
            self.ctx._check(self.lib.tracker_stage_time(self.h, i, C.byref(ms), C.byref(n)), "tracker_stage_time")
            out[name] = 1e3 * ms.value / max(1, n.value)
        return out

    def track_sequence_native(self, kf, d_frames, motion, opts, shuffle_levels, shuffle_fine, passes=1, poses_true=None):
        """ptam_bench_track_sequence (ptam_hip_bench.h): the frames tracked in order, closed loop, from one native host thread
        -> (seconds, stats dict)"""
        n = len(d_frames)
        raw = lambda d: d.p.value if isinstance(d, DevBuf) else int(d)
        dis = (C.c_void_p * n)(*[raw(d) for d in d_frames])
        sl = np.ascontiguousarray(shuffle_levels, dtype=np.int32)
        sf = np.ascontiguousarray(shuffle_fine, dtype=np.int32)
        pt = None if poses_true is None else np.ascontiguousarray(poses_true, dtype=np.float64).reshape(n, 12)
        secs = C.c_double()
        st = np.zeros(8)
        self.ctx._check(self.lib.bench_track_sequence(self.h, kf.h, n, dis, C.byref(motion), _ptr(opts) if opts is not None else None,
                                                      _ptr(sl), _ptr(sf), int(passes), _ptr(pt), C.byref(secs), _pd(st)), "bench_track_sequence")
        keys = ("frames", "searched", "templates_reused", "measurements", "frames_did_coarse", "frames_tried_coarse",
                "max_position_error_m", "frames_below_50_measurements")
        return secs.value, dict(zip(keys, (float(x) for x in st)))

    @staticmethod
    def TrackFramesBatch(trackers, kfs, d_frames, poses, opts=None):
        """ptam_track_map_frames_batch: one chain of launches for len(trackers) frames -> array of TRACKMAP_RESULT_DT"""
        k = len(trackers)
        raw = lambda h: h.value if hasattr(h, "value") else int(h)
        trs = (C.c_void_p * k)(*[raw(t.h) for t in trackers])
        kf_ = (C.c_void_p * k)(*[raw(f.h) for f in kfs])
        dis = (C.c_void_p * k)(*[raw(d.p if isinstance(d, DevBuf) else d) for d in d_frames])
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(k, 12))
        res = np.zeros(k, dtype=TRACKMAP_RESULT_DT)
        t0 = trackers[0]
        t0.ctx._check(t0.lib.track_map_frames_batch(k, trs, kf_, dis, _pd(poses), _ptr(opts) if opts is not None else None, _ptr(res)),
                      "track_map_frames_batch")
        return res

    def iteration_set(self):
        n = C.c_int()
        self.ctx._check(self.lib.tracker_read_iteration_set(self.h, None, 0, C.byref(n)), "tracker_read_iteration_set")
        out = np.zeros(n.value, dtype=TRACKMAP_MEAS_DT)
        if n.value:
            self.ctx._check(self.lib.tracker_read_iteration_set(self.h, _ptr(out), n.value, C.byref(n)), "tracker_read_iteration_set")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.tracker_destroy(self.h)
            self.h = None


class FrameTracker:
    """The per-frame fine stage of Tracker::TrackMap with everything resident on the device: SearchForPoints over a
    batch of queries (src/Tracker.cc:867-912: FindPatchCoarse, found patches become pose measurements) followed by the
    ten pose iterations (:613-643).  Only the 96-byte pose comes back to the host."""

    def __init__(self, ctx, capacity):
        self.ctx, self.lib, self.cap = ctx, ctx.lib, int(capacity)
        self.d_res = DevBuf(ctx, self.cap * PATCH_RESULT_DT.itemsize)
        self.d_meas = DevBuf(ctx, self.cap * POSE_MEAS_DT.itemsize)
        self.d_src = DevBuf(ctx, self.cap * 4)
        self.d_flags = DevBuf(ctx, self.cap * 4)
        self.d_count = DevBuf(ctx, 32)      # count | manMeasFound[4]
        self.d_pose = DevBuf(ctx, 96)

    def search_and_update(self, kf, n, d_queries, d_templates, d_world, world_stride, pose, opts=None, d_subpix=None):
        """d_* are DevBuf (or raw device pointers).  pose: the prediction, or None to go on from the resident pose of
        the previous frame.  Returns the refined pose (12 doubles)."""
        raw = lambda b: b.p if isinstance(b, DevBuf) else b
        assert 1 <= n <= self.cap
        chk, lib, h = self.ctx._check, self.lib, self.ctx.h
        pose_in = None if pose is None else np.array(pose, dtype=np.float64).reshape(12).copy()
        opts = opts or self.ctx.gn_opts()
        chk(lib.find_patch_coarse_batch_dev(h, kf.h, n, raw(d_queries), raw(d_templates), self.d_res.p), "find_patch_coarse_dev")
        cnt = C.c_void_p(self.d_count.p.value)
        lvl = C.c_void_p(self.d_count.p.value + 8)
        chk(lib.gather_pose_meas_dev(h, n, raw(d_queries), self.d_res.p, raw(d_subpix) if d_subpix is not None else None,
                                     raw(d_world), world_stride, self.d_meas.p, self.d_src.p, cnt, lvl), "gather_pose_meas_dev")
        out = np.zeros(12)
        chk(lib.pose_gn_dev_counted(h, n, cnt, self.d_meas.p, None, self.d_pose.p, C.byref(opts), self.d_flags.p, None,
                                    _pd(pose_in) if pose_in is not None else None, _pd(out)), "pose_gn_dev_counted")
        return out

    def last_measurements(self):
        """(count, measurements, source query indices, outlier flags, found-per-level) of the last frame"""
        head = self.d_count.download(np.int32, 8)
        n = int(head[0])
        return (n, self.d_meas.download(POSE_MEAS_DT, n), self.d_src.download(np.int32, n), self.d_flags.download(np.int32, n),
                head[2:6].copy())

    def close(self):
        for b in (self.d_res, self.d_meas, self.d_src, self.d_flags, self.d_count, self.d_pose):
            b.free()


class Bundle:
    """Bundle (include/Bundle.h:106-152)."""

    def __init__(self, ctx, **opts):
        self.ctx, self.lib = ctx, ctx.lib
        o = BaOpts()
        self.lib.ba_opts_default(C.byref(o))
        for k, v in opts.items():
            setattr(o, k, v)
        self.opts = o
        h = C.c_void_p()
        ctx._check(self.lib.ba_create(ctx.h, C.byref(o), C.byref(h)), "ba_create")
        self.h = h
        self._keep = []

    def AddCamera(self, pose, fixed):
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        return self.ctx._check(self.lib.ba_add_camera(self.h, _pd(pose), int(fixed)), "ba_add_camera")

    def AddPoint(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(3)
        return self.ctx._check(self.lib.ba_add_point(self.h, _pd(pos)), "ba_add_point")

    def AddMeas(self, cam, point, found, sigma_sq):
        found = np.ascontiguousarray(found, dtype=np.float64).reshape(2)
        self.ctx._check(self.lib.ba_add_meas(self.h, int(cam), int(point), _pd(found), float(sigma_sq)),
                        "ba_add_meas")

    def add_problem(self, poses, fixed, points, cam_idx, pt_idx, found, sigma_sq):
        """bulk marshalling: same ids as calling AddCamera/AddPoint/AddMeas in array order"""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 12)
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        cam_idx = np.ascontiguousarray(cam_idx, dtype=np.int32)
        pt_idx = np.ascontiguousarray(pt_idx, dtype=np.int32)
        found = np.ascontiguousarray(found, dtype=np.float64).reshape(-1, 2)
        sigma_sq = np.ascontiguousarray(sigma_sq, dtype=np.float64)
        c = self.ctx._check
        c(self.lib.ba_add_cameras(self.h, len(poses), _ptr(poses), _ptr(fixed)), "ba_add_cameras")
        c(self.lib.ba_add_points(self.h, len(points), _ptr(points)), "ba_add_points")
        c(self.lib.ba_add_measurements(self.h, len(cam_idx), _ptr(cam_idx), _ptr(pt_idx), _ptr(found),
                                       _ptr(sigma_sq)), "ba_add_measurements")

    def Compute(self, abort=None):
        """-> mnAccepted (or -1).  abort: optional np.uint8 array of length 1 polled by the library."""
        acc = C.c_int()
        self.ctx._check(self.lib.ba_compute(self.h, _ptr(abort), C.byref(acc)), "ba_compute")
        return acc.value

    def Converged(self):
        return bool(self.lib.ba_converged(self.h))

    def counts(self):
        v = [C.c_int() for _ in range(4)]
        self.ctx._check(self.lib.ba_counts(self.h, *[C.byref(x) for x in v]), "ba_counts")
        return tuple(x.value for x in v)   # cams, free cams, points, meas

    def GetPoint(self, n):
        out = np.zeros(3)
        self.ctx._check(self.lib.ba_get_point(self.h, n, _pd(out)), "ba_get_point")
        return out

    def GetCamera(self, n):
        out = np.zeros(12)
        self.ctx._check(self.lib.ba_get_camera(self.h, n, _pd(out)), "ba_get_camera")
        return out

    def get_all(self):
        nc, _, npt, _ = self.counts()
        poses, pts = np.zeros((nc, 12)), np.zeros((npt, 3))
        self.ctx._check(self.lib.ba_get_all(self.h, _ptr(poses), _ptr(pts)), "ba_get_all")
        return poses, pts

    def GetOutlierMeasurements(self):
        n = self.lib.ba_get_outliers(self.h, None, 0)
        out = np.zeros((max(n, 1), 2), dtype=np.int32)
        self.lib.ba_get_outliers(self.h, _ptr(out), n)
        return out[:n]   # (point, camera) pairs

    def trials(self):
        n = self.lib.ba_get_trials(self.h, None, 0)
        out = np.zeros(max(n, 1), dtype=BA_TRIAL_DT)
        self.lib.ba_get_trials(self.h, _ptr(out), n)
        return out[:n]

    def set_comm(self, rank, world, fn, user=None):
        self._keep.append(fn)
        self.ctx._check(self.lib.ba_set_comm(self.h, rank, world, fn, user), "ba_set_comm")

    # profiling hooks (HIP library only)
    def set_profiling(self, on=True):
        self.ctx._check(self.lib.ba_set_profiling(self.h, int(on)), "ba_set_profiling")

    def kernel_times(self):
        out = {}
        for k, name in enumerate(_abi.KERNEL_NAMES):
            ms, n = C.c_double(), C.c_int()
            self.ctx._check(self.lib.ba_kernel_time(self.h, k, C.byref(ms), C.byref(n)), "ba_kernel_time")
            out[name] = (ms.value, n.value)
        return out

    def solve_fallbacks(self):
        """trials repeated with the launch-per-column camera solve after the persistent one gave up a wait (0 normally)"""
        return int(self.lib.ba_solve_fallbacks(self.h)) if self.lib.has("ba_solve_fallbacks") else 0

    def prepare(self):
        self.ctx._check(self.lib.ba_prepare(self.h), "ba_prepare")

    def duplicates_refused(self):
        """measurements of the last prepare that had a twin (same point, same camera): such a bundle is refused"""
        return int(self.lib.ba_duplicates_refused(self.h))

    def debug_lists(self, which, dtype):
        """one of the index structures ptam_ba_prepare built on the device (include/ptam_hip_bench.h: PTAM_BL_*), flat"""
        n = self.ctx._check(self.lib.ba_debug_lists(self.h, which, None, 0), "ba_debug_lists")
        out = np.zeros(max(n, 1), dtype=np.uint8)
        self.ctx._check(self.lib.ba_debug_lists(self.h, which, _ptr(out), n), "ba_debug_lists")
        return out[:n].view(dtype)

    def bench_jacobian(self, reps):
        ms, by = C.c_double(), C.c_double()
        self.ctx._check(self.lib.ba_bench_jacobian(self.h, reps, C.byref(ms), C.byref(by)), "ba_bench_jacobian")
        return ms.value, by.value

    @staticmethod
    def bench_jacobian_rotating(bundles, reps):
        """K7 round-robin over copies of one problem (Infinity-Cache-cold launches) -> average ms per launch"""
        arr = (C.c_void_p * len(bundles))(*[b.h for b in bundles])
        ms = C.c_double()
        bundles[0].ctx._check(bundles[0].lib.ba_bench_jacobian_rotating(arr, len(bundles), reps, C.byref(ms)), "ba_bench_jacobian_rotating")
        return ms.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.ba_destroy(self.h)
            self.h = None
