"""CPU: the tracker's motion model (src/Tracker.cc:1008-1056) and SE3 ln / exp — known answers against scipy's matrix
logarithm, the product's host code (libptam_hip.so: plain scalar functions, no device) against the oracle's restatement —
and the moving-camera sequence of synth.py tracked closed loop by the oracle (Tracker::TrackFrame, :94, :134-137)."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.linalg

from ptam_cg_amd import _abi, host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hostlib():
    """the product library bound WITHOUT a context: only its host-side scalar entry points are called here"""
    return _abi.bind(C.CDLL(os.path.join(ROOT, "ptam_cg_amd", "csrc", "libptam_hip.so")), "ptam_")


def _exp(lib, mu):
    out = np.zeros(12)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    lib.se3_exp(mu.ctypes.data, out.ctypes.data)
    return out


def _ln(lib, pose):
    out = np.zeros(6)
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    lib.se3_ln(pose.ctypes.data, out.ctypes.data)
    return out


@pytest.mark.parametrize("theta", [0.0, 1e-7, 1e-5, 5e-4, 2e-3, 0.1, 0.7, 0.8, 2.0, 2.3, 2.4, 3.0, 3.14, np.pi - 1e-9])
def test_se3_ln_inverts_exp_and_matches_logm(oracle, hostlib, theta):
    """all three ranges of TooN's SO3::ln (asin below pi/4, acos up to 3 pi/4, symmetric part beyond) and both ranges of the
    translation correction of SE3::ln"""
    rng = np.random.default_rng(int(theta * 1000) + 3)
    for _ in range(6):
        w = rng.normal(size=3)
        w *= theta / np.linalg.norm(w)
        mu = np.concatenate([rng.normal(size=3), w])
        T = _exp(oracle, mu)
        back = _ln(oracle, T)
        # (TooN switches between series and closed forms at theta = 1e-5 / 1e-3 / 1e-4: the forms differ by O(theta^2 t) there)
        assert np.abs(back - mu).max() < 1e-10 * max(1.0, 1.0 / max(np.pi - theta, 1e-3)), (theta, back, mu)
        assert np.abs(_ln(hostlib, T) - back).max() <= 1e-14            # product host code == oracle restatement (hipcc contracts a*b+c)
        assert np.abs(_exp(hostlib, mu) - T).max() <= 1e-14
        if theta < 3.1:
            M = np.eye(4)
            M[:3, :3], M[:3, 3] = T[:9].reshape(3, 3), T[9:]
            L = scipy.linalg.logm(M).real
            ref = np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])
            assert np.abs(back - ref).max() < 1e-7, (theta, back, ref)


def _result(pose, depth_n=500, depth_mean=1.4):
    r = np.zeros(1, dtype=host.TRACKMAP_RESULT_DT)
    r["pose"] = pose
    r["depth_n"] = depth_n
    r["depth_sum"] = depth_mean * depth_n
    r["depth_sum_sq"] = (depth_mean ** 2 + 0.01) * depth_n
    return r


@pytest.mark.parametrize("const_vel", [1, 0])
def test_motion_model_product_equals_oracle_and_follows_a_constant_twist(oracle, hostlib, const_vel):
    """a camera that moves by the same twist every frame: after one update the constant-velocity model predicts the next pose
    exactly (src/Tracker.cc:1029 with :1042); the decaying model (:1045-1046) lags behind.  Product and oracle agree."""
    twist = np.array([0.01, -0.004, 0.002, 0.003, -0.02, 0.015])
    step = _exp(oracle, twist)
    pose = synth.look_at([0.3, -0.2, 1.5], [0, 0, 0])
    models = []
    for lib in (oracle, hostlib):
        m = _abi.MotionModel()
        lib.motion_reset(C.byref(m), host._pd(np.ascontiguousarray(pose)))
        m.use_constant_velocity = const_vel
        models.append(m)
    assert models[0].scene_depth_mean == 1.0 and models[0].coarse_min_velocity == 0.006
    cur = pose.copy()
    for k in range(4):
        cur = synth.se3_mul(step, cur)
        for lib, m in zip((oracle, hostlib), models):
            lib.motion_predict(C.byref(m))
            pred = np.array(m.pose)
            lib.motion_update(C.byref(m), host._ptr(_result(cur, depth_n=500 if k else 10)))
            if lib is oracle:
                keep = pred
        a, b = models
        assert np.abs(np.array(a.velocity) - np.array(b.velocity)).max() <= 1e-16 and np.array_equal(np.array(a.pose), np.array(b.pose))
        assert abs(a.msd_scaled_velocity - b.msd_scaled_velocity) <= 1e-16 and a.scene_depth_mean == b.scene_depth_mean
        if const_vel:
            assert np.abs(np.array(a.velocity) - twist).max() < 1e-13
            if k >= 1:
                assert np.abs(keep - cur).max() < 1e-13                 # predicted = where the camera went
        elif k >= 1:
            assert 1e-4 < np.abs(keep - cur).max()
    # the depth of the scene only follows frames that found more than 20 points (:692)
    assert abs(models[0].scene_depth_mean - 1.4) < 1e-12
    v = np.array(models[0].velocity)
    want = np.sqrt((v[:3] ** 2).sum() / 1.4 ** 2 + (v[3:] ** 2).sum())
    assert abs(models[0].msd_scaled_velocity - want) < 1e-15


@pytest.fixture(scope="module")
def sequence():
    return synth.make_tracking_frames(16)


def test_sequence_is_deterministic_and_textured(sequence):
    frames, poses, kim, kpose = sequence
    again = synth.render_plane_view(synth.AtanCam(), poses[3], synth.make_plane_texture(),
                                    None)
    assert frames.shape == (16, 480, 640) and frames.dtype == np.uint8 and kim.shape == (480, 640)
    assert np.abs(frames[3].astype(int) - again.astype(int)).max() <= 2          # the same view up to its sensor noise
    assert frames.std() > 30 and not np.array_equal(frames[0], frames[1])
    for p in list(poses) + [kpose]:
        R = p[:9].reshape(3, 3)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12


def test_oracle_tracks_the_moving_camera_closed_loop(oracle, sequence):
    """Tracker::TrackFrame's tracking branch frame after frame: the pose stays on the true trajectory, most patches are found,
    part of the templates is re-warped (the warps move past the 0.07 limit of src/PatchFinder.cc:103-111) and part is kept,
    the coarse stage switches on once the model has picked up the velocity (:505)"""
    frames, poses, kim, kpose = sequence
    ctx = host.Context(lib=oracle)
    kf0 = host.KeyFrame(ctx).MakeKeyFrame_Lite(kim)
    m = synth.make_sequence_map([kf0.level(l) for l in range(4)], kpose, counts=(400, 200, 70, 40))
    tr = host.Tracker(ctx, len(m["world"]))
    tr.set_map(m["world"], m["pixel_right_w"], m["pixel_down_w"], kf0, m["src_level"], m["center"])
    kf = host.KeyFrame(ctx)
    mm = tr.motion_model(poses[0])
    reused = searched = coarse = 0
    for k in range(len(frames)):
        tr.set_shuffle(m["shuffle_levels"], m["shuffle_fine"])
        r = tr.TrackFrameMoving(kf, frames[k].ctypes.data, mm, tr.opts())
        assert np.abs(r["pose"] - poses[k]).max() < 3e-3, k
        assert np.array_equal(np.array(mm.pose), r["pose"])
        n = int(r["n_coarse"] + r["n_top"] + r["n_fine"])
        assert r["n_meas"] > 0.7 * n
        if k:
            reused, searched, coarse = reused + int(r["templates_reused"]), searched + n, coarse + int(r["did_coarse"])
        else:
            assert r["did_coarse"] == 0 and r["templates_reused"] == 0       # no velocity yet, fresh finders
    assert 0.2 * searched < reused < 0.95 * searched
    assert coarse >= len(frames) - 3
    tr.close()
