"""ctypes view of include/ptam_hip.h: struct layouts and prototypes.

`bind(lib, prefix)` attaches argtypes/restypes for every entry point that the given shared library
exports under `prefix` ("ptam_" for libptam_hip.so).  The binding is prefix-generic so that the
test-suite can drive its CPU checker through the very same host classes; the package itself only
ever loads libptam_hip.so (see _lib.py).
"""
import ctypes as C

LEVELS = 4
MAX_SSD = 8 * 8 * 500
HALFSAMPLE_R, HALFSAMPLE_T = 0, 1
EST_TUKEY, EST_CAUCHY, EST_HUBER = 0, 1, 2
K_PROJECT, K_SELECT, K_JACOBIAN, K_VINV, K_SCHUR, K_SOLVE, K_UPDATE, K_EXCHANGE, K_COUNT = range(9)
KERNEL_NAMES = ["project", "select", "jacobian", "vinv", "schur", "solve", "update", "exchange"]
TRACK_STAGE_NAMES = ["pyramid_pvs", "fast_detect", "compact_select", "search_coarse", "gather_coarse", "pose_coarse",
                     "search_fine", "gather_fine", "pose_fine"]


class Int2(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32)]


class CamParams(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("w", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class PatchQuery(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("level", C.c_int32), ("range", C.c_uint32)]


class PatchResult(C.Structure):
    _fields_ = [("found", C.c_int32), ("best_ssd", C.c_int32), ("best_x", C.c_int32),
                ("best_y", C.c_int32), ("n_scored", C.c_int32), ("pad_", C.c_int32),
                ("pos", C.c_double * 2)]


class Projection(C.Structure):
    _fields_ = [("cam", C.c_double * 3), ("image", C.c_double * 2), ("derivs", C.c_double * 4),
                ("in_image", C.c_int32), ("pad_", C.c_int32)]


class SubpixQuery(C.Structure):
    _fields_ = [("coarse_pos", C.c_double * 2), ("level", C.c_int32), ("max_its", C.c_int32)]


class SubpixResult(C.Structure):
    _fields_ = [("converged", C.c_int32), ("iterations", C.c_int32), ("pos", C.c_double * 2), ("mean_diff", C.c_double)]


class TemplateQuery(C.Structure):
    _fields_ = [("src_kf", C.c_void_p), ("src_level", C.c_int32), ("search_level", C.c_int32), ("center_x", C.c_int32),
                ("center_y", C.c_int32), ("warp_inverse", C.c_double * 4)]


class TemplateResult(C.Structure):
    _fields_ = [("bad", C.c_int32), ("n_outside", C.c_int32), ("sum", C.c_int32), ("sum_sq", C.c_int32), ("m2", C.c_double * 4)]


class EpipolarQuery(C.Structure):
    _fields_ = [("level_x", C.c_int32), ("level_y", C.c_int32), ("normal", C.c_double * 2), ("norm_dist", C.c_double),
                ("along", C.c_double * 2), ("min_len", C.c_double), ("max_len", C.c_double), ("max_dist_sq", C.c_double)]


class EpipolarResult(C.Structure):
    _fields_ = [("best", C.c_int32), ("best_zmssd", C.c_int32), ("n_scored", C.c_int32), ("template_bad", C.c_int32)]


class PvsPoint(C.Structure):
    _fields_ = [("world", C.c_double * 3), ("pixel_right_w", C.c_double * 3), ("pixel_down_w", C.c_double * 3)]


class PoseMeas(C.Structure):
    _fields_ = [("world", C.c_double * 3), ("found", C.c_double * 2), ("sqrt_inv_noise", C.c_double)]


class GnOpts(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("nonlinear_mask", C.c_uint32),
                ("override_after", C.c_int32), ("override_sigma_sq", C.c_double),
                ("mark_outliers_iter", C.c_int32), ("estimator", C.c_int32), ("prior", C.c_double)]


class PoseUpdateMeas(C.Structure):
    _fields_ = [("found", C.c_double * 2), ("image", C.c_double * 2), ("sqrt_inv_noise", C.c_double),
                ("jac", C.c_double * 12)]


class BaOpts(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("update_sq_conv_limit", C.c_double),
                ("min_sigma", C.c_double), ("estimator", C.c_int32), ("verbose", C.c_int32),
                ("deterministic", C.c_int32), ("pad_", C.c_int32)]


class MotionModel(C.Structure):
    """ptam_motion_model (src/Tracker.cc:1008-1056)"""
    _fields_ = [("pose", C.c_double * 12), ("start_pose", C.c_double * 12), ("velocity", C.c_double * 6),
                ("msd_scaled_velocity", C.c_double), ("scene_depth_mean", C.c_double), ("scene_depth_sigma", C.c_double),
                ("coarse_min_velocity", C.c_double), ("use_constant_velocity", C.c_int32), ("disable_coarse", C.c_int32),
                ("just_recovered", C.c_int32), ("pad_", C.c_int32)]


class BaTrial(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("sigma_sq", C.c_double), ("err_old", C.c_double),
                ("err_new", C.c_double), ("sum_sq_update", C.c_double), ("n_bad", C.c_int32),
                ("accepted", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_void_p)

_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
_pd, _pi, _pu8 = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
_ppv = C.POINTER(C.c_void_p)

# name -> (restype, argtypes).  Names are given WITHOUT the prefix.
PROTOTYPES = {
    "last_error": (C.c_char_p, []),
    "device_count": (_i, [C.POINTER(_i)]),
    "ctx_create": (_i, [C.POINTER(CamParams), _i, _ppv]),
    "ctx_destroy": (_i, [_vp]),
    "ctx_set_halfsample": (_i, [_vp, _i]),
    "ctx_sync": (_i, [_vp]),
    "ctx_stream": (_vp, [_vp]),
    "ctx_camera_constants": (_i, [_vp, _pd]),
    "dev_alloc": (_i, [_vp, C.c_size_t, _ppv]),
    "dev_free": (_i, [_vp, _vp]),
    "dev_upload": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "dev_download": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "kf_create": (_i, [_vp, _i, _i, _ppv]),
    "kf_destroy": (_i, [_vp]),
    "make_keyframe_lite": (_i, [_vp, _vp, _vp, _i]),
    "make_keyframe_lite_dev": (_i, [_vp, _vp, _vp]),
    "kf_clone": (_i, [_vp, _vp, _ppv]),
    "kf_level_info": (_i, [_vp, _vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "kf_read_level": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "make_keyframe_rest": (_i, [_vp, _vp]),
    "kf_rest_info": (_i, [_vp, _vp, _i, C.POINTER(_i)]),
    "kf_read_rest": (_i, [_vp, _vp, _i, _vp, _vp]),
    "find_patch_coarse_batch": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "find_patch_coarse_batch_dev": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "zmssd_at_points": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "project_points": (_i, [_vp, _i, _vp, _pd, _vp]),
    "subpix_batch": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "make_templates_batch": (_i, [_vp, _i, _vp, _vp, _vp]),
    "kf_implane_corners": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "epipolar_search_batch": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "ctx_one_pixel_dist": (_i, [_vp, _pd]),
    "ctx_cache_hazards": (_i, [_vp, C.POINTER(C.c_longlong)]),
    "track_pvs": (_i, [_vp, _i, _vp, _pd, _vp, _vp]),
    "gn_opts_default": (None, [C.POINTER(GnOpts)]),
    "pose_gn": (_i, [_vp, _i, _vp, _vp, _pd, C.POINTER(GnOpts), _vp, _vp]),
    "pose_gn_dev": (_i, [_vp, _i, _vp, _vp, _vp, C.POINTER(GnOpts), _vp, _vp]),
    "pose_gn_dev_counted": (_i, [_vp, _i, _vp, _vp, _vp, _vp, C.POINTER(GnOpts), _vp, _vp, _pd, _pd]),
    "gather_pose_meas_dev": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "calc_pose_update": (_i, [_vp, _i, _vp, _d, _i, _d, _pd, _vp]),
    "ba_opts_default": (None, [C.POINTER(BaOpts)]),
    "ba_create": (_i, [_vp, C.POINTER(BaOpts), _ppv]),
    "ba_destroy": (_i, [_vp]),
    "ba_add_camera": (_i, [_vp, _pd, _i]),
    "ba_add_point": (_i, [_vp, _pd]),
    "ba_add_meas": (_i, [_vp, _i, _i, _pd, _d]),
    "ba_add_cameras": (_i, [_vp, _i, _vp, _vp]),
    "ba_add_points": (_i, [_vp, _i, _vp]),
    "ba_add_measurements": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "ba_compute": (_i, [_vp, _vp, C.POINTER(_i)]),
    "ba_converged": (_i, [_vp]),
    "ba_get_point": (_i, [_vp, _i, _pd]),
    "ba_get_camera": (_i, [_vp, _i, _pd]),
    "ba_get_all": (_i, [_vp, _vp, _vp]),
    "ba_get_outliers": (_i, [_vp, _vp, _i]),
    "ba_get_trials": (_i, [_vp, _vp, _i]),
    "ba_counts": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "ba_solve_fallbacks": (_i, [_vp]),
    "ba_duplicates_refused": (_i, [_vp]),
    "ba_set_profiling": (_i, [_vp, _i]),
    "ba_kernel_time": (_i, [_vp, _i, _pd, C.POINTER(_i)]),
    "ba_prepare": (_i, [_vp]),
    "ba_bench_jacobian": (_i, [_vp, _i, _pd, _pd]),
    "reproject_points": (_i, [_vp, _i, _vp, _pd, _vp]),
    "refind_batch": (_i, [_vp, _vp, _pd, _i, _vp, _vp, _vp]),
    "refinder_create": (_i, [_vp, C.POINTER(_vp)]),
    "refinder_destroy": (_i, [_vp]),
    "refind_pairs": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "pose_gn_state": (_i, [_vp, _i, _vp, _vp, _pd, _vp, _vp, _vp, _vp]),
    "trackmap_opts_default": (None, [_vp]),
    "tracker_create": (_i, [_vp, _i, _ppv]),
    "tracker_destroy": (_i, [_vp]),
    "tracker_set_map": (_i, [_vp, _i, _vp, _vp]),
    "tracker_update_map": (_i, [_vp, _i, _vp, _vp, _vp]),
    "tracker_set_shuffle": (_i, [_vp, _vp, _vp]),
    "track_map": (_i, [_vp, _vp, _pd, _vp, _vp]),
    "track_map_frame": (_i, [_vp, _vp, _vp, _pd, _vp, _vp]),
    "motion_reset": (None, [C.POINTER(MotionModel), _pd]),
    "motion_predict": (None, [C.POINTER(MotionModel)]),
    "motion_update": (None, [C.POINTER(MotionModel), _vp]),
    "se3_exp": (None, [_vp, _vp]),
    "se3_ln": (None, [_vp, _vp]),
    "track_frame": (_i, [_vp, _vp, _vp, C.POINTER(MotionModel), _vp, _vp]),
    "bench_track_sequence": (_i, [_vp, _vp, _i, _vp, C.POINTER(MotionModel), _vp, _vp, _vp, _i, _vp, _pd, _pd]),
    "tracker_set_profiling": (_i, [_vp, _i]),
    "tracker_stage_time": (_i, [_vp, _i, _pd, C.POINTER(_i)]),
    "bench_track_frames": (_i, [_i, _vp, _vp, _vp, _pd, _vp, _vp, _vp, _i, _pd]),
    "track_map_frames_batch": (_i, [_i, _vp, _vp, _vp, _pd, _vp, _vp]),
    "bench_track_batch": (_i, [_i, _vp, _vp, _vp, _pd, _vp, _vp, _vp, _i, _i, _pd]),
    "tracker_read_iteration_set": (_i, [_vp, _vp, _i, C.POINTER(_i)]),
    "ba_bench_jacobian_rotating": (_i, [_vp, _i, _i, _pd]),
    "ba_schur_index_map": (_i, [_i, _vp, _i]),
    "ba_debug_lists": (_i, [_vp, _i, _vp, C.c_size_t]),
    "ba_set_comm": (_i, [_vp, _i, _i, ALLREDUCE_FN, _vp]),
    "rccl_unique_id": (_i, [_vp]),
    "rccl_create": (_i, [_vp, _vp, _i, _i, _ppv]),
    "rccl_destroy": (_i, [_vp]),
    "rccl_allreduce_f64": (_i, [_vp, _pd, C.c_size_t, _vp]),
}

# every symbol include/ptam_hip.h declares (checked by the CPU test-suite against the built .so)
DECLARED = sorted(PROTOTYPES)


class Bound:
    """Attribute access to `<prefix><name>` functions of one shared library."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        self.missing = []
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(lib, prefix + name)
            except AttributeError:
                self.missing.append(name)
                continue
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)

    def has(self, name):
        return hasattr(self, name)


def bind(lib, prefix):
    return Bound(lib, prefix)
