"""Test-side loader and driver of the extended-precision referee (oracle/libptam_referee.so: the oracle's Bundle with every
double an x87 long double).  Array arguments of its ptamo_ba_* entry points are numpy.longdouble; the ABI structs keep their
layout.  Used by tests/ and tests/tools only."""
import ctypes as C
import os
import subprocess

import numpy as np

from ptam_cg_amd import _abi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "libptam_referee.so")
_lib = None


def load_referee():
    global _lib
    if _lib is None:
        src = [os.path.join(ORACLE_DIR, f) for f in ("referee.cc", "ptam_oracle.cc")]
        if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(f) for f in src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "libptam_referee.so"])
        assert np.finfo(np.longdouble).nmant == 63, "numpy.longdouble is not the x87 extended format here"
        _lib = C.CDLL(SO)
        for name in ("ptamo_ctx_create", "ptamo_ba_create", "ptamo_ba_add_cameras", "ptamo_ba_add_points", "ptamo_ba_add_measurements",
                     "ptamo_ba_compute", "ptamo_ba_get_all", "ptamo_ba_get_trials", "ptamo_ba_get_outliers", "ptamo_ba_destroy",
                     "ptamo_ctx_destroy", "ptamo_ba_converged"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def run_ba(prob, camera=host.DEFAULT_CAMERA, size=(640, 480), **opts):
    """util.run_ba through the referee: the same marshalling, the state and the trial log back as float64 (the trial log's
    struct holds doubles; poses and points are rounded from extended precision)"""
    lib = load_referee()
    cam = _abi.CamParams(*camera, size[0], size[1])
    ctx = C.c_void_p()
    assert lib.ptamo_ctx_create(C.byref(cam), 0, C.byref(ctx)) == 0
    o = _abi.BaOpts()
    C.CDLL(os.path.join(ORACLE_DIR, "libptam_oracle.so")).ptamo_ba_opts_default(C.byref(o))   # (a plain struct of the header: either library fills it)
    for k, v in opts.items():
        setattr(o, k, v)
    ba = C.c_void_p()
    assert lib.ptamo_ba_create(ctx, C.byref(o), C.byref(ba)) == 0
    ld = np.longdouble
    poses = np.ascontiguousarray(prob["poses"], dtype=ld)
    fixed = np.ascontiguousarray(prob["fixed"], dtype=np.uint8)
    pts = np.ascontiguousarray(prob["points"], dtype=ld)
    cam_i = np.ascontiguousarray(prob["cam_idx"], dtype=np.int32)
    pt_i = np.ascontiguousarray(prob["pt_idx"], dtype=np.int32)
    found = np.ascontiguousarray(prob["found"], dtype=ld)
    sig = np.ascontiguousarray(prob["sigma_sq"], dtype=ld)
    assert lib.ptamo_ba_add_cameras(ba, len(poses), _p(poses), _p(fixed)) == 0
    assert lib.ptamo_ba_add_points(ba, len(pts), _p(pts)) == 0
    assert lib.ptamo_ba_add_measurements(ba, len(cam_i), _p(cam_i), _p(pt_i), _p(found), _p(sig)) == 0
    acc = C.c_int()
    assert lib.ptamo_ba_compute(ba, None, C.byref(acc)) == 0
    out_poses = np.zeros((len(poses), 12), dtype=ld)
    out_pts = np.zeros((len(pts), 3), dtype=ld)
    assert lib.ptamo_ba_get_all(ba, _p(out_poses), _p(out_pts)) == 0
    tr = np.zeros(4096, dtype=host.BA_TRIAL_DT)
    n = lib.ptamo_ba_get_trials(ba, _p(tr), len(tr))
    cap = 2 * len(cam_i) + 2
    ol = np.zeros(cap, dtype=np.int32)
    no = lib.ptamo_ba_get_outliers(ba, _p(ol), cap // 2)
    res = {"accepted": acc.value, "converged": bool(lib.ptamo_ba_converged(ba)), "trials": tr[:n].copy(),
           "poses": out_poses.astype(np.float64), "points": out_pts.astype(np.float64), "outliers": ol[:2 * no].reshape(-1, 2).copy()}
    lib.ptamo_ba_destroy(ba)
    lib.ptamo_ctx_destroy(ctx)
    return res
