"""-m gpu: parity of the HIP path (through the C ABI) against the CPU oracle on identical inputs.
Integers (pyramid pixels, corner lists, LUTs, ZMSSD, best index) bit-exact; fp64 pose / bundle
results within the tolerances written in each test (north_star: 1e-6 relative on the post-LM error)."""
import ctypes as C

import numpy as np
import pytest

from ptam_cg_amd import _abi, host, synth
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [_abi.HALFSAMPLE_R, _abi.HALFSAMPLE_T])
@pytest.mark.parametrize("case", ["synthetic", "shifted", "noise", "odd", "flat", "tiny"])
def test_keyframe_lite_bit_exact(hip, oracle, variant, case):
    rng = np.random.default_rng(7)
    if case == "synthetic":
        im = synth.make_frame()
    elif case == "shifted":
        im = synth.make_frame_pair()[1]
    elif case == "noise":
        im = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    elif case == "odd":
        im = rng.integers(0, 256, (77, 101), dtype=np.uint8)   # ragged: halves drop the last row/col
        im[20:60, 30:80] = 200
    elif case == "flat":
        im = np.full((480, 640), 77, np.uint8)                  # no corners at all: empty lists, zero LUTs
    else:
        im = rng.integers(0, 256, (8, 8), dtype=np.uint8)       # minimum size: level 3 is 1x1
    a = util.keyframe_levels(hip, im, variant)
    b = util.keyframe_levels(oracle, im, variant)
    util.assert_levels_equal(a, b)
    if case == "synthetic":
        assert sum(len(x["corners"]) for x in a) > 1000


def test_keyframe_clone_and_reuse(hip, oracle):
    a, b = synth.make_frame_pair()
    ctx = host.Context(lib=hip)
    kf = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
    cl = kf.clone()                    # Level::operator= deep copy
    kf.MakeKeyFrame_Lite(b)            # re-use the handle for the next frame
    util.assert_levels_equal([cl.level(l) for l in range(4)], util.keyframe_levels(oracle, a))
    util.assert_levels_equal([kf.level(l) for l in range(4)], util.keyframe_levels(oracle, b))


def _patch_case(lib, search_range, n=1000):
    a, b = synth.make_frame_pair()
    ctx = host.Context(lib=lib)
    kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
    kfb = host.KeyFrame(ctx).MakeKeyFrame_Lite(b)
    q, t = synth.make_patch_queries([kfa.level(l) for l in range(4)], n=n, search_range=search_range)
    return ctx, kfb, q, t


@pytest.mark.parametrize("search_range", [10, 30, 120])
def test_find_patch_coarse_bit_exact(hip, oracle, search_range):
    ctx_h, kf_h, q, t = _patch_case(hip, search_range)
    ctx_o, kf_o, q2, t2 = _patch_case(oracle, search_range)
    assert np.array_equal(q, q2) and np.array_equal(t, t2)
    # edge cases: bad template, off-image predictions, zero range, border positions
    q = q.copy()
    q[0]["level"] = -1
    q[1]["x"], q[1]["y"] = -50, -50
    q[2]["x"], q[2]["y"] = 5000, 100
    q[3]["range"] = 0
    q[4]["x"], q[4]["y"] = 0, 0
    q[5]["x"], q[5]["y"] = 639, 479
    rh = host.PatchFinder(ctx_h).FindPatchCoarse(kf_h, q, t)
    ro = host.PatchFinder(ctx_o).FindPatchCoarse(kf_o, q, t)
    for f in ("found", "best_ssd", "best_x", "best_y", "n_scored"):
        assert np.array_equal(rh[f], ro[f]), f
    assert np.array_equal(rh["pos"], ro["pos"])
    assert rh["found"].sum() > 0.8 * len(q) or search_range == 120


def test_zmssd_at_points_bit_exact(hip, oracle):
    im = synth.make_frame()
    rng = np.random.default_rng(3)
    pts = np.column_stack([rng.integers(-2, 84, 500), rng.integers(-2, 64, 500)]).astype(np.int32)
    for level in range(4):
        ctx_h, ctx_o = host.Context(lib=hip), host.Context(lib=oracle)
        kh = host.KeyFrame(ctx_h).MakeKeyFrame_Lite(im)
        ko = host.KeyFrame(ctx_o).MakeKeyFrame_Lite(im)
        tmpl = rng.integers(0, 256, 64, dtype=np.uint8)
        sh = host.PatchFinder(ctx_h).ZMSSDAtPoint(kh, level, pts, tmpl)
        so = host.PatchFinder(ctx_o).ZMSSDAtPoint(ko, level, pts, tmpl)
        assert np.array_equal(sh, so)
        if level == 3:
            assert (so == _abi.MAX_SSD + 1).any() and (so != _abi.MAX_SSD + 1).any()


def test_project_points(hip, oracle):
    pc = synth.make_pose_case()
    rng = np.random.default_rng(5)
    world = np.vstack([pc["world"], rng.uniform(-3, 3, (500, 3))])   # many invisible / behind-camera points
    ph = host.Context(lib=hip).project_points(world, pc["init_pose"])
    po = host.Context(lib=oracle).project_points(world, pc["init_pose"])
    assert np.array_equal(ph["in_image"], po["in_image"])
    for f in ("cam", "image", "derivs"):
        assert np.allclose(ph[f], po[f], rtol=1e-12, atol=1e-9), f
    assert 0 < ph["in_image"].sum() < len(world)


@pytest.mark.parametrize("stage", ["fine", "coarse"])
def test_pose_gn(hip, oracle, stage):
    pc = synth.make_pose_case()
    ch, co = host.Context(lib=hip), host.Context(lib=oracle)
    kw = {} if stage == "fine" else dict(nonlinear_mask=0x3FF, override_sigma_sq=1.0, mark_outliers_iter=-1)
    ph, fh, uh = ch.pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"], ch.gn_opts(**kw))
    po, fo, uo = co.pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"], co.gn_opts(**kw))
    assert np.allclose(ph, po, rtol=0, atol=1e-10)          # final pose
    assert np.allclose(uh, uo, rtol=1e-6, atol=1e-12)       # every iteration's 6-vector update
    assert np.array_equal(fh, fo)                           # M-estimator outlier marks
    if stage == "fine":
        assert fh.sum() >= 0.8 * pc["is_outlier"].sum()
    assert np.abs(ph - pc["true_pose"]).max() < 5e-3


def _pose_case_with_points_leaving_the_camera_model(n, seed):
    """make_pose_case + 16 map points that project half a pixel inside the image corner farthest from the principal point — where
    the camera model's radius ends (src/ATANCamera.cc:56-63): a pose update of a pixel in the wrong direction takes them outside,
    TrackerData::Project bails out before the model (include/Tracker.h:73-80) and ProjectAndDerivs (:89-94) would read stale derivatives"""
    pc = synth.make_pose_case(n=n, seed=seed)
    cam = synth.AtanCam()
    rng = np.random.default_rng(seed)
    px = rng.uniform(0.05, 0.45, (16, 2))                      # pixel (0, 0) is the far corner of the default camera (cx, cy > 0.5)
    d = np.column_stack([cam.unproject(px), np.ones(16)])      # rays in the camera frame of the INITIAL pose
    R, t = pc["init_pose"][:9].reshape(3, 3), pc["init_pose"][9:]
    o, dw = -R.T @ t, d @ R                                    # camera centre and ray directions in the world
    lam = -o[2] / dw[:, 2]                                     # onto the plane z = 0
    extra = o + lam[:, None] * dw
    ok0, _ = cam.visible(pc["init_pose"], extra)
    assert ok0.all()
    _, im_true = cam.visible(pc["true_pose"], extra)
    return {"world": np.vstack([pc["world"], extra]), "found": np.vstack([pc["found"], im_true]),
            "sqrt_inv_noise": np.concatenate([pc["sqrt_inv_noise"], np.ones(16)]), "init_pose": pc["init_pose"]}


@pytest.mark.parametrize("n", [40, 600, 1500])   # one wave | 512 x 2 in registers | the general kernel
def test_pose_gn_counts_the_projections_that_leave_the_camera_model(hip, oracle, n):
    """The deviation of DESIGN section 2 is observable: ptam_ctx_cache_hazards counts the re-projections of found measurements that bail
    out before the camera model — the same events the checker counts — and the pose loop's result is the checker's."""
    ch, co = host.Context(lib=hip), host.Context(lib=oracle)
    seen = 0
    for seed in range(40, 60):
        pc = _pose_case_with_points_leaving_the_camera_model(n, seed)
        h0, o0 = ch.cache_hazards(), co.cache_hazards()
        ph, fh, uh = ch.pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"])
        po, fo, uo = co.pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"])
        dh, do = ch.cache_hazards() - h0, co.cache_hazards() - o0
        assert dh == do, (seed, dh, do)
        assert np.allclose(ph, po, rtol=0, atol=1e-10) and np.array_equal(fh, fo)
        seen += do
        if seen >= 3:
            break
    assert seen >= 3   # (the construction does produce the event)


def test_pose_gn_with_massive_ties(hip, oracle):
    """12 distinct measurements, each repeated 80 times: the squared errors tie in runs of 80, so the order-statistic
    select cannot finish from one histogram bin and has to take its general radix path"""
    pc = synth.make_pose_case(n=12)
    rep = lambda a: np.repeat(a, 80, axis=0)
    ch, co = host.Context(lib=hip), host.Context(lib=oracle)
    ph, fh, uh = ch.pose_gn(rep(pc["world"]), rep(pc["found"]), rep(pc["sqrt_inv_noise"]), pc["init_pose"])
    po, fo, uo = co.pose_gn(rep(pc["world"]), rep(pc["found"]), rep(pc["sqrt_inv_noise"]), pc["init_pose"])
    assert np.allclose(ph, po, rtol=0, atol=1e-10) and np.array_equal(fh, fo)
    assert np.allclose(uh, uo, rtol=1e-6, atol=1e-12)


def test_pose_gn_device_resident_equals_host_entry(hip):
    """ptam_pose_gn_dev on resident buffers = ptam_pose_gn on host buffers, bit for bit (same kernel)"""
    pc = synth.make_pose_case()
    ctx = host.Context(lib=hip)
    ref, flags_ref, _ = ctx.pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"])
    n = len(pc["world"])
    meas = np.zeros(n, dtype=host.POSE_MEAS_DT)
    meas["world"], meas["found"], meas["sqrt_inv_noise"] = pc["world"], pc["found"], pc["sqrt_inv_noise"]
    d_m, d_p, d_f = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for ptr, nbytes in ((d_m, meas.nbytes), (d_p, 96), (d_f, 4 * n)):
        ctx._check(hip.dev_alloc(ctx.h, nbytes, C.byref(ptr)), "alloc")
    pose = pc["init_pose"].copy()
    ctx._check(hip.dev_upload(ctx.h, d_m, meas.ctypes.data, meas.nbytes), "up")
    ctx._check(hip.dev_upload(ctx.h, d_p, pose.ctypes.data, 96), "up")
    opts = ctx.gn_opts()
    ctx._check(hip.pose_gn_dev(ctx.h, n, d_m, None, d_p, C.byref(opts), d_f, None), "pose_gn_dev")
    out, flags = np.zeros(12), np.zeros(n, dtype=np.int32)
    ctx._check(hip.dev_download(ctx.h, out.ctypes.data, d_p, 96), "down")
    ctx._check(hip.dev_download(ctx.h, flags.ctypes.data, d_f, 4 * n), "down")
    assert np.array_equal(out, ref) and np.array_equal(flags, flags_ref)
    for ptr in (d_m, d_p, d_f):
        hip.dev_free(ctx.h, ptr)


def test_device_resident_frame_equals_host_path(hip):
    """SearchForPoints' bookkeeping + pose solve with nothing leaving the device (ptam_gather_pose_meas_dev,
    ptam_pose_gn_dev_counted) against the same steps through the host entries: src/Tracker.cc:883-909 restated in numpy
    for the measurement list, then bit-identical poses / outlier flags (same kernels, same list)."""
    ctx = host.Context(lib=hip)
    a, b = synth.make_frame_pair()
    kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
    q, t = synth.make_patch_queries([kfa.level(l) for l in range(4)], n=1000)
    q = q.copy()
    q["level"][::37] = -1                       # TemplateBad: never searched, never a measurement
    kfb = host.KeyFrame(ctx).MakeKeyFrame_Lite(b)
    pc = synth.make_pose_case()
    rng = np.random.default_rng(3)
    world = pc["world"][rng.integers(0, len(pc["world"]), len(q))]
    pvs = np.zeros(len(q), dtype=[("world", "<f8", (3,)), ("r", "<f8", (3,)), ("d", "<f8", (3,))])   # ptam_pvs_point layout
    pvs["world"] = world
    pf = host.PatchFinder(ctx)
    res = pf.FindPatchCoarse(kfb, q, t)
    sub = pf.SubPix(kfb, res["pos"], np.where(res["found"] != 0, q["level"], -1), t)
    ft = host.FrameTracker(ctx, len(q))
    d_q, d_t, d_w, d_sub = host.DevBuf(ctx, q), host.DevBuf(ctx, t), host.DevBuf(ctx, pvs), host.DevBuf(ctx, sub)
    for use_sub in (False, True):
        keep = (q["level"] >= 0) & (res["found"] != 0)
        if use_sub:
            keep &= sub["converged"] != 0
        idx = np.nonzero(keep)[0]
        found = (sub if use_sub else res)["pos"][idx]
        sn = 1.0 / (1 << q["level"][idx]).astype(np.float64)
        ref_pose, ref_flags, _ = ctx.pose_gn(world[idx], found, sn, pc["init_pose"])
        pose = ft.search_and_update(kfb, len(q), d_q, d_t, d_w, pvs.itemsize, pc["init_pose"], d_subpix=d_sub if use_sub else None)
        n, meas, src, flags, per_level = ft.last_measurements()
        assert n == len(idx) and n > 300 and np.array_equal(src, idx)
        assert np.array_equal(meas["world"], world[idx]) and np.array_equal(meas["found"], found)
        assert np.array_equal(meas["sqrt_inv_noise"], sn)
        assert np.array_equal(per_level, np.bincount(q["level"][idx], minlength=4))
        assert np.array_equal(pose, ref_pose) and np.array_equal(flags, ref_flags)
    # nothing found: zero measurements, pose unchanged (src/Tracker.cc:955-956)
    q0 = q.copy()
    q0["level"] = -1
    d_q0 = host.DevBuf(ctx, q0)
    pose = ft.search_and_update(kfb, len(q), d_q0, d_t, d_w, pvs.itemsize, pc["init_pose"])
    assert ft.last_measurements()[0] == 0 and np.array_equal(pose, pc["init_pose"])
    for buf in (d_q, d_t, d_w, d_sub, d_q0):
        buf.free()
    ft.close()


def test_bundle_memory_is_reused_across_bundles(hip):
    """A context keeps a released bundle's device block and mailbox for the next bundle (MapMaker builds a new Bundle per
    adjustment).  Bundles of changing size created, run and destroyed in a row — two alive at a time — must give
    what a fresh context gives (stale contents, stale mailbox sequence numbers and too-small blocks would all show)."""
    cases = [dict(n_cams=6, n_pts=120, seed=1), dict(n_cams=12, n_pts=700, seed=2), dict(n_cams=5, n_pts=60, seed=3),
             dict(n_cams=12, n_pts=700, seed=2), dict(n_cams=20, n_pts=1500, seed=4, window=6), dict(n_cams=6, n_pts=120, seed=1)]
    fresh = []
    for c in cases:
        fresh.append(util.run_ba(hip, synth.make_ba_problem(**c)))
    ctx = host.Context(lib=hip)
    prev = None
    for c, ref in zip(cases, fresh):
        prob = synth.make_ba_problem(**c)
        ba = synth.load_into(host.Bundle(ctx), prob)
        acc = ba.Compute()
        poses, pts = ba.get_all()
        got = {"accepted": acc, "converged": ba.Converged(), "trials": ba.trials(), "poses": poses, "points": pts,
               "outliers": ba.GetOutlierMeasurements()}
        util.assert_ba_equal(got, ref, rel=1e-6)   # (not bit for bit: K7's LDS atomics are unordered)
        if prev is not None:
            prev.close()   # (released only now: the next bundle finds one cached block, the one after that two)
        prev = ba
    prev.close()
    ctx.close()


def test_pose_gn_entry_state_and_empty(hip, oracle):
    pc = synth.make_pose_case(n=300)
    ch, co = host.Context(lib=hip), host.Context(lib=oracle)
    # entry state = projections at a slightly different pose (what TrackMap leaves after a coarse stage)
    other = synth.se3_mul(synth.se3_exp(np.array([1e-3, -2e-3, 5e-4, 1e-3, 0, -1e-3])), pc["init_pose"])
    entry = co.project_points(pc["world"], other)
    ph, fh, uh = ch.pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"], entry=entry)
    po, fo, uo = co.pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"], entry=entry)
    assert np.allclose(ph, po, rtol=0, atol=1e-10) and np.array_equal(fh, fo)
    # empty set: zero update, pose unchanged (src/Tracker.cc:955-956)
    p0, _, u0 = ch.pose_gn(np.zeros((0, 3)), np.zeros((0, 2)), np.zeros(0), pc["init_pose"])
    assert np.array_equal(p0, pc["init_pose"]) and not u0.any()


@pytest.mark.parametrize("est", [_abi.EST_TUKEY, _abi.EST_CAUCHY, _abi.EST_HUBER])
@pytest.mark.parametrize("override", [0.0, 16.0])
def test_calc_pose_update(hip, oracle, est, override):
    rng = np.random.default_rng(11)
    n = 777
    found = rng.uniform(0, 640, (n, 2))
    image = found + rng.normal(0, 1.5, (n, 2))
    image[::20] += 40
    s = 1.0 / 2.0 ** rng.integers(0, 4, n)
    jac = rng.normal(0, 300, (n, 12))
    mh, fh = host.Context(lib=hip).calc_pose_update(found, image, s, jac, override, est)
    mo, fo = host.Context(lib=oracle).calc_pose_update(found, image, s, jac, override, est)
    assert np.allclose(mh, mo, rtol=1e-9, atol=1e-15)
    assert np.array_equal(fh, fo)


BA_CASES = {
    "toy_8x50": dict(n_cams=8, n_pts=50, seed=1),
    "local_20x300": dict(n_cams=20, n_pts=300, seed=synth.SEED_BA_LOCAL),
    "banded_40x400": dict(n_cams=40, n_pts=400, seed=3, window=10),
    "two_fixed": dict(n_cams=12, n_pts=120, seed=4, n_fixed=2),
    "config4_20x3000": dict(n_cams=20, n_pts=3000, seed=synth.SEED_BA_LOCAL),
    "wide_80x60": dict(n_cams=80, n_pts=60, seed=6),          # > 64 measurements per point: block variant of K7
    # the camera solve as ONE persistent launch (ldlt_chain.inc: 6 .. 12 block rows): the smallest system that takes it, a banded
    # one whose row workers hold only part of their row, and the largest (12 blocks: the last rows' workers lag behind the chain)
    "chain_33x500": dict(n_cams=33, n_pts=500, seed=41),
    "chain_58x700_w20": dict(n_cams=58, n_pts=700, seed=42, window=20),
    "chain_64x900": dict(n_cams=64, n_pts=900, seed=43),
    # 19 block rows with a band of ~9: too many for a dense worker's LDS, not banded enough for two chains -> one persistent chain
    # whose row workers hold only their band; and 24 block rows of band ~5: two persistent chains with four columns each
    "chain_100x900_w40": dict(n_cams=100, n_pts=900, seed=44, window=40),
    "twisted_128x1200_w24_chains": dict(n_cams=128, n_pts=1200, seed=45, window=24),
    # long banded trajectories: camera system of 23 / 35 blocks with a 2- / 3-block band -> two-ended LDL^T (solve.hip)
    "twisted_120x1500_w8": dict(n_cams=120, n_pts=1500, seed=21, window=8),
    "twisted_181x2500_w12_f3": dict(n_cams=181, n_pts=2500, seed=22, window=12, n_fixed=3),
}


# BASELINE.json's full sizes and the shapes that select the other K7 forms: a few lambda trials each so that the
# single-thread oracle stays within seconds
BA_BIG_CASES = {
    "headline_50x5000": dict(n_cams=50, n_pts=5000, seed=synth.SEED_BA_HEADLINE),                 # one chunk per wave
    "loop256_60x7000": dict(n_cams=60, n_pts=7000, seed=11),                                      # looping K7, 256 threads
    "loop512_200x26000_w16": dict(n_cams=200, n_pts=26000, seed=synth.SEED_BA_GLOBAL, window=16),   # LDS-bound: 512 threads
    # BASELINE.json configs[4] at its full size on ONE device (M ~ 0.8 M, camera system 1194 x 1194, band 4 blocks);
    # three trials: the oracle's dense 1194^3/3 factorisation and its C x P scans take seconds each
    "config5_200x50000_w16": dict(n_cams=200, n_pts=50000, seed=synth.SEED_BA_GLOBAL, window=16),
}
BA_BIG_TRIALS = {"config5_200x50000_w16": 3}


@pytest.mark.parametrize("case", list(BA_BIG_CASES))
def test_bundle_full_size_trial_by_trial(hip, oracle, case):
    prob = synth.make_ba_problem(**BA_BIG_CASES[case])
    k = BA_BIG_TRIALS.get(case, 4)
    rh = util.run_ba(hip, prob, max_iterations=k)
    ro = util.run_ba(oracle, prob, max_iterations=k)
    util.assert_ba_equal(rh, ro, rel=1e-6)
    assert len(rh["trials"]) == k and rh["accepted"] > 0
    # size-independent properties: every accepted trial lowers the robust error, rejected ones do not move the state
    t = rh["trials"]
    assert all(x["err_new"] < x["err_old"] for x in t if x["accepted"])
    assert all(not (x["err_new"] < x["err_old"]) for x in t if not x["accepted"])


@pytest.mark.parametrize("case", list(BA_CASES))
def test_bundle_trial_by_trial(hip, oracle, case):
    prob = synth.make_ba_problem(**BA_CASES[case])
    rh = util.run_ba(hip, prob)
    ro = util.run_ba(oracle, prob)
    util.assert_ba_equal(rh, ro, rel=1e-6)
    assert rh["accepted"] > 0
    # the adjuster actually improved the map (a long open chain of cameras with 8-12 views per point drifts as a whole in the
    # free gauge directions: there the robust error falls while the distance to the generating positions need not)
    if not case.startswith("twisted"):
        assert np.abs(rh["points"] - prob["points_true"]).mean() < np.abs(prob["points"] - prob["points_true"]).mean()
    else:
        assert rh["trials"]["err_new"][-1] < rh["trials"]["err_old"][0]


# src/Bundle.cc:46-93 bounds neither the cameras of a map nor the measurements of a point; until round 3 the device path
# refused a point seen by more than 256 cameras, ~600 free cameras, and 255 fixed cameras between two free ones of a
# Schur tile.  Each shape below crosses one of those former limits (VERDICT r2 missing 2).
BA_UNBOUNDED_CASES = {
    # every point is measured by 300 cameras: chunks of ONE point walked 256 measurements at a time (pass 1, the point update)
    "long_points_300x40": (dict(n_cams=300, n_pts=40, seed=31), 3),
    # 300 FIXED cameras in front of 9 free ones, every point seen by all 309: the Schur work lists address a tile's cameras by
    # 8-bit offsets inside the point's row, which only works because the free cameras' measurements come first
    "many_fixed_309x60": (dict(n_cams=309, n_pts=60, seed=32, n_fixed=300), 4),
    # 639 free cameras: K7's camera partials (138 KB) + poses (61 KB) exceed a workgroup's LDS -> partials and poses in
    # global memory; the backward substitution's vectors (6 x 3840 doubles) exceed it too
    "many_cameras_640x1500_w6": (dict(n_cams=640, n_pts=1500, seed=33, window=6), 1),   # (one trial: the oracle factors 3834 x 3834 densely, ~40 s)
}


@pytest.mark.parametrize("case", list(BA_UNBOUNDED_CASES))
def test_bundle_sizes_the_reference_does_not_bound(hip, oracle, case):
    kw, k = BA_UNBOUNDED_CASES[case]
    prob = synth.make_ba_problem(**kw)
    rh = util.run_ba(hip, prob, max_iterations=k)
    ro = util.run_ba(oracle, prob, max_iterations=k)
    util.assert_ba_equal(rh, ro, rel=1e-6)
    assert 1 <= len(rh["trials"]) <= k and rh["accepted"] > 0


# 420 cameras: the one-chunk-per-wave K7 form at 1024 threads no longer fits the CU's LDS (partials + poses + 16 transposition
# buffers), and the deterministic mode has no 512-thread instantiation: it must take the looping 256-thread form
BA_CASES_DET_EXTRA = {"many_cams_420x700_w10": dict(n_cams=420, n_pts=700, seed=52, window=10)}


@pytest.mark.parametrize("case", ["local_20x300", "banded_40x400", "two_fixed", "many_cams_420x700_w10"])
def test_bundle_deterministic_mode_is_bit_reproducible(hip, oracle, case):
    """ptam_ba_opts.deterministic (VERDICT r2 missing 5, SURVEY section 7 "offer a deterministic two-index mode"): the camera sums
    of pass 2 in a fixed order.  Five runs must agree to the last bit in every trial's numbers, the poses and the points, and
    the mode must pass the same parity check against the oracle as the default one."""
    big = case in BA_CASES_DET_EXTRA
    kw = dict(max_iterations=4) if big else {}      # (the oracle's dense 2514-row factorisation takes seconds per trial)
    prob = synth.make_ba_problem(**(BA_CASES_DET_EXTRA[case] if big else BA_CASES[case]))
    runs = [util.run_ba(hip, prob, deterministic=1, **kw) for _ in range(3 if big else 5)]
    for r in runs[1:]:
        for k in runs[0]["trials"].dtype.names:
            assert np.array_equal(r["trials"][k], runs[0]["trials"][k], equal_nan=True), k
        assert np.array_equal(r["poses"], runs[0]["poses"]) and np.array_equal(r["points"], runs[0]["points"])
        assert np.array_equal(r["outliers"], runs[0]["outliers"])
    util.assert_ba_equal(runs[0], util.run_ba(oracle, prob, **kw), rel=1e-6)


def test_bundle_deterministic_mode_headline_size(hip):
    """the same at 50 keyframes x 5000 points (the looping K7 form with stored Jacobians, 250 000 measurements): two runs of
    four trials, bit-identical"""
    prob = synth.make_ba_problem(**BA_BIG_CASES["headline_50x5000"])
    a = util.run_ba(hip, prob, max_iterations=4, deterministic=1)
    b = util.run_ba(hip, prob, max_iterations=4, deterministic=1)
    for k in a["trials"].dtype.names:
        assert np.array_equal(a["trials"][k], b["trials"][k]), k
    assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"])
    c = util.run_ba(hip, prob, max_iterations=4)            # and the default mode agrees with it to the usual tolerance
    assert np.allclose(a["trials"]["err_new"], c["trials"]["err_new"], rtol=1e-9)


def test_bundle_headline_full_length_deterministic_against_oracle(hip, oracle):
    """the headline problem at its FULL length — Bundle.MaxIterations = 20 trials — in deterministic
    mode against the oracle, trial by trial (lambda, accept / reject, bad counts, sigma^2, errors, outliers, poses, points).
    (a) as PTAM runs it (convergence test on): the whole trajectory; (b) as bench.py runs it (convergence limit 0: exactly 20
    trials, the second half at the noise floor, where a trial comes back with new == current error to the last bit or 4e-12
    better): the whole trajectory up to the first trial whose improvement is below 1e-9 of the error — from there on the
    discrete outcome is decided by the last bits of two different summation orders — and every error to 1e-6 throughout."""
    prob = synth.make_ba_problem(**BA_BIG_CASES["headline_50x5000"])
    rh = util.run_ba(hip, prob, max_iterations=20, deterministic=1)
    ro = util.run_ba(oracle, prob, max_iterations=20)
    util.assert_ba_equal(rh, ro, rel=1e-6)
    assert rh["accepted"] >= 5
    rh0 = util.run_ba(hip, prob, max_iterations=20, update_sq_conv_limit=0.0, deterministic=1)
    ro0 = util.run_ba(oracle, prob, max_iterations=20, update_sq_conv_limit=0.0)
    th, to = rh0["trials"], ro0["trials"]
    assert len(th) == len(to) == 20
    floor = next((i for i, x in enumerate(to) if abs(x["err_new"] - x["err_old"]) <= 1e-9 * abs(x["err_old"])), 20)
    assert floor >= 5, floor
    for i in range(floor):
        assert th[i]["lambda"] == to[i]["lambda"] and th[i]["accepted"] == to[i]["accepted"] and th[i]["n_bad"] == to[i]["n_bad"], i
    for i in range(20):
        for k in ("err_old", "err_new"):
            assert abs(th[i][k] - to[i][k]) <= 1e-6 * abs(to[i][k]), (i, k, th[i][k], to[i][k])


@pytest.mark.parametrize("case", ["config4_20x3000", "chain_33x500", "chain_58x700_w20", "headline_50x5000"])
def test_backward_substitution_inside_the_launch_equals_the_separate_kernel(hip, case):
    """round 5: for camera systems whose band is at most 9 blocks the persistent solve's right-hand-side workgroup substitutes
    backwards itself (ldlt_chain.inc, wave roles); PTAM_LDLT_SEPARATE_BACKWARD=1 puts ldlt_backward_kernel behind the launch
    as before.  The two sum in different (both fixed) orders: the same discrete trajectory, every trial's numbers to 1e-9, the
    state to 1e-9 — and, deterministic mode, the in-launch form to the last bit against itself."""
    big = case in BA_BIG_CASES
    kw = dict(max_iterations=6) if big else {}
    c = BA_BIG_CASES[case] if big else BA_CASES[case]
    a = util.run_ba_subprocess(c, opts=dict(deterministic=1, **kw))
    b = util.run_ba_subprocess(c, env={"PTAM_LDLT_SEPARATE_BACKWARD": "1"}, opts=dict(deterministic=1, **kw))
    a2 = util.run_ba_subprocess(c, opts=dict(deterministic=1, **kw))
    assert a["solve_fallbacks"] == 0 and b["solve_fallbacks"] == 0
    util.assert_ba_equal(a, b, rel=1e-9, abs_state=1e-9)
    for k in a["trials"].dtype.names:
        assert np.array_equal(a["trials"][k], a2["trials"][k], equal_nan=True), k
    assert np.array_equal(a["poses"], a2["poses"]) and np.array_equal(a["points"], a2["points"])


def _fuzz_cases(n=18, seed=2024):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        n_cams = int(rng.integers(3, 71))
        n_pts = int(rng.integers(6, 500))
        window = None if rng.random() < 0.4 else int(rng.integers(2, max(3, n_cams)))
        out.append(dict(n_cams=n_cams, n_pts=n_pts, seed=1000 + i, window=window, n_fixed=int(rng.integers(1, min(4, n_cams))),
                        outlier_frac=float(rng.choice([0.0, 0.02, 0.1])), pt_noise=float(rng.choice([0.002, 0.01, 0.03]))))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: f"{c['n_cams']}x{c['n_pts']}w{c['window']}f{c['n_fixed']}")
def test_bundle_random_shapes(hip, oracle, case):
    """random camera counts (tile / block padding of the Schur and LDL^T kernels), covisibility windows (absent-camera
    zero blocks, banded systems), several fixed cameras, outlier rates: trial by trial against the oracle"""
    prob = synth.make_ba_problem(**case)
    if len(prob["cam_idx"]) == 0:
        pytest.skip("no measurement survived the visibility test")
    est = [_abi.EST_TUKEY, _abi.EST_CAUCHY, _abi.EST_HUBER][case["seed"] % 3]
    util.assert_ba_equal(util.run_ba(hip, prob, estimator=est), util.run_ba(oracle, prob, estimator=est), rel=1e-6)


@pytest.mark.parametrize("n_cams,fixed", [(30, (0, 3, 4, 11, 17)), (21, (0, 9, 10, 20)), (13, (2,)), (27, (0, 1, 8, 16, 24, 26))],
                         ids=lambda v: str(v).replace(" ", ""))
def test_bundle_fixed_cameras_anywhere(hip, oracle, n_cams, fixed):
    """fixed cameras scattered through the camera order (the reference fixes whichever keyframes lie outside the
    adjusted set, src/MapMaker.cc:791-824): their measurements sit BETWEEN those of a Schur tile's free cameras, so the
    per-entry camera-slot tables of K8 must skip them; the free-camera counts 25 / 17 / 12 / 21 leave last tiles of
    1 / 1 / 4 / 5 cameras (fragment modes 1, 1, 2, 2 of ba_schur.inc) beside full ones"""
    prob = synth.make_ba_problem(n_cams, 260, 77 + n_cams, window=None if n_cams < 25 else 14)
    prob["fixed"][:] = 0
    prob["fixed"][list(fixed)] = 1
    util.assert_ba_equal(util.run_ba(hip, prob), util.run_ba(oracle, prob), rel=1e-6)


@pytest.mark.parametrize("est", [_abi.EST_CAUCHY, _abi.EST_HUBER])
def test_bundle_other_estimators(hip, oracle, est):
    prob = synth.make_ba_problem(10, 150, 9)
    util.assert_ba_equal(util.run_ba(hip, prob, estimator=est), util.run_ba(oracle, prob, estimator=est), rel=1e-6)


def test_bundle_noise_free_is_fixed_point(hip):
    prob = synth.make_ba_problem(6, 80, 2, outlier_frac=0.0)
    cam = synth.AtanCam()
    # exact measurements from the true geometry, start at the truth
    prob["poses"], prob["points"] = prob["poses_true"].copy(), prob["points_true"].copy()
    f = np.zeros_like(prob["found"])
    for i, (c, p) in enumerate(zip(prob["cam_idx"], prob["pt_idx"])):
        _, im = cam.visible(prob["poses_true"][c], prob["points_true"][p:p + 1])
        f[i] = im[0]
    prob["found"] = f
    r = util.run_ba(hip, prob)
    assert r["trials"]["err_old"].max() < 1e-9
    assert np.allclose(r["poses"], prob["poses_true"], atol=1e-9)
    assert np.allclose(r["points"], prob["points_true"], atol=1e-9)
    assert len(r["outliers"]) == 0


def test_bundle_tiny_errors_and_degenerate_systems(hip, oracle):
    """(a) errors far below 2^-24 (the first-level bin of the select is then the clamped end bin: a full-width radix
    pass — a 64-bit shift by 64 once returned garbage there); (b) one point seen by six cameras: five free cameras
    constrained by twelve equations, a singular camera system that only the damping keeps solvable — both sides must
    walk the same trials and must terminate"""
    prob = synth.make_ba_problem(6, 80, 2, outlier_frac=0.0)
    cam = synth.AtanCam()
    f = np.zeros_like(prob["found"])
    for i, (c, p) in enumerate(zip(prob["cam_idx"], prob["pt_idx"])):
        _, im = cam.visible(prob["poses_true"][c], prob["points_true"][p:p + 1])
        f[i] = im[0]
    prob["found"] = f + np.random.default_rng(3).normal(0, 1e-6, f.shape)      # e^2 ~ 1e-12
    prob["poses"], prob["points"] = prob["poses_true"].copy(), prob["points_true"].copy()
    rh, ro = util.run_ba(hip, prob, max_iterations=4), util.run_ba(oracle, prob, max_iterations=4)
    util.assert_ba_equal(rh, ro, rel=1e-6)
    assert rh["trials"]["sigma_sq"].max() < 1.0          # = MinTukeySigma^2, not a garbage order statistic
    p = synth.make_ba_problem(6, 40, 3)
    keep = p["pt_idx"] == 0
    q = {k: (v[keep] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in p.items()}
    q["points"], q["points_true"] = p["points"][:1], p["points_true"][:1]
    rh, ro = util.run_ba(hip, q, max_iterations=6), util.run_ba(oracle, q, max_iterations=6)
    assert len(rh["trials"]) == len(ro["trials"]) and np.array_equal(rh["trials"]["accepted"], ro["trials"]["accepted"])
    assert np.allclose(rh["trials"]["sigma_sq"], ro["trials"]["sigma_sq"], rtol=1e-6)


def test_bundle_abort_and_limits(hip):
    prob = synth.make_ba_problem(6, 60, 5)
    ctx = host.Context(lib=hip)
    ba = synth.load_into(host.Bundle(ctx), prob)
    flag = np.ones(1, dtype=np.uint8)                       # abort already requested: nothing accepted
    assert ba.Compute(abort=flag) == 0 and len(ba.trials()) == 0 and not ba.Converged()
    assert ba.Compute() > 0                                 # and the same object still works afterwards
    ba2 = synth.load_into(host.Bundle(ctx, max_iterations=3), prob)
    ba2.Compute()
    assert len(ba2.trials()) == 3                           # Bundle.MaxIterations counts lambda trials
    with pytest.raises(host.PtamError):
        ba2.AddMeas(99, 0, [1.0, 2.0], 1.0)                 # unknown camera id


def test_subpix_matches_oracle(hip, oracle):
    ctx_h, kf_h, q, t = _patch_case(hip, 10, n=600)
    ctx_o, kf_o, _, _ = _patch_case(oracle, 10, n=600)
    r = host.PatchFinder(ctx_o).FindPatchCoarse(kf_o, q, t)
    ok = np.flatnonzero(r["found"])
    pos, lv = r["pos"][ok], q["level"][ok].copy()
    lv[0] = -1                                                   # skipped query
    pos[1] = (2.0, 2.0)                                          # border: fails on the first iteration
    sh = host.PatchFinder(ctx_h).SubPix(kf_h, pos, lv, t[ok], 8)
    so = host.PatchFinder(ctx_o).SubPix(kf_o, pos, lv, t[ok], 8)
    assert np.array_equal(sh["converged"], so["converged"]) and np.array_equal(sh["iterations"], so["iterations"])
    assert np.allclose(sh["pos"], so["pos"], rtol=0, atol=1e-9)
    assert np.allclose(sh["mean_diff"], so["mean_diff"], rtol=0, atol=1e-9)
    assert so["converged"].mean() > 0.8


def test_make_template_coarse_cont_bit_exact(hip, oracle):
    """MakeTemplateCoarseCont: warped template bytes, outside counts, sums and the warp matrix, full-size frame,
    templates then fed to the coarse search on both sides"""
    a, b = synth.make_frame_pair()
    tc = synth.make_template_cases((a.shape[1], a.shape[0]), n=2000)
    out = {}
    for name, lib in (("hip", hip), ("oracle", oracle)):
        ctx = host.Context(lib=lib)
        kfa, kfb = host.KeyFrame(ctx).MakeKeyFrame_Lite(a), host.KeyFrame(ctx).MakeKeyFrame_Lite(b)
        pf = host.PatchFinder(ctx)
        tm, r = pf.MakeTemplateCoarseCont(kfa, tc["src_level"], tc["center"], tc["search_level"], tc["warp_inverse"])
        q = np.zeros(len(tm), dtype=host.PATCH_QUERY_DT)
        lvl0 = tc["center"].astype(np.int64) << tc["src_level"][:, None]
        q["x"], q["y"], q["range"] = lvl0[:, 0] + 3, lvl0[:, 1] - 2, 12
        q["level"] = np.where(r["bad"] != 0, -1, tc["search_level"])
        out[name] = (tm, r, pf.FindPatchCoarse(kfb, q, tm))
    (th, rh, fh), (to, ro, fo) = out["hip"], out["oracle"]
    assert np.array_equal(th, to)
    for f in ("bad", "n_outside", "sum", "sum_sq", "m2"):
        assert np.array_equal(rh[f], ro[f]), f
    for f in ("found", "best_ssd", "best_x", "best_y", "n_scored"):
        assert np.array_equal(fh[f], fo[f]), f
    assert 0 < np.count_nonzero(ro["n_outside"]) < len(to) // 2 and ro["bad"][0] == 1


@pytest.mark.parametrize("level", [0, 1, 3])
def test_epipolar_corner_scan_matches_oracle(hip, oracle, level):
    """AddPointEpipolar's scan on the full-size frame pair: in-plane corner table (tan() differs by an ulp between
    libm and the device), and per candidate the best corner index / ZMSSD / scored count, all exact"""
    a, b = synth.make_frame_pair()
    cam = synth.AtanCam()
    out = {}
    for name, lib in (("hip", hip), ("oracle", oracle)):
        ctx = host.Context(lib=lib)
        kfa, kfb = host.KeyFrame(ctx).MakeKeyFrame_Lite(a), host.KeyFrame(ctx).MakeKeyFrame_Lite(b)
        opd = ctx.one_pixel_dist()
        q = synth.make_epipolar_queries(cam, kfa.level(level)["corners"], level, opd, n=1500, seed=77 + level)
        out[name] = (opd, kfb.implane_corners(level), host.PatchFinder(ctx).EpipolarSearch(kfa, kfb, level, q))
    (oh, ih, rh), (oo, io, ro) = out["hip"], out["oracle"]
    assert abs(oh - oo) <= 1e-16 and np.allclose(ih, io, rtol=1e-14, atol=1e-16)
    for f in ("best", "best_zmssd", "n_scored", "template_bad"):
        assert np.array_equal(rh[f], ro[f]), f
    assert np.count_nonzero(ro["best"] >= 0) > 700 and ro["n_scored"].max() > 5


def test_track_pvs_matches_oracle(hip, oracle):
    pv = synth.make_pvs_case()
    rh, ch = host.Context(lib=hip).track_pvs(pv["world"], pv["pixel_right_w"], pv["pixel_down_w"], pv["pose"])
    ro, co = host.Context(lib=oracle).track_pvs(pv["world"], pv["pixel_right_w"], pv["pixel_down_w"], pv["pose"])
    assert np.array_equal(rh["level"], ro["level"]) and np.array_equal(ch, co)
    assert np.array_equal(rh["proj"]["in_image"], ro["proj"]["in_image"])
    for f in ("cam", "image", "derivs"):
        assert np.allclose(rh["proj"][f], ro["proj"][f], rtol=1e-12, atol=1e-9), f
    assert np.allclose(rh["warp_inverse"], ro["warp_inverse"], rtol=1e-11, atol=1e-12)
    assert (co > 0).all() and (ro["level"] == -1).sum() > 0


def test_two_host_threads_two_contexts(hip, oracle):
    """The reference enters this path from two OS threads (tracker: keyframes / search / pose,
    mapmaker: bundle adjustment, src/MapMaker.cc:57).  One context per thread, run concurrently;
    both must reproduce the single-threaded results."""
    import threading
    a, b = synth.make_frame_pair()
    prob = synth.make_ba_problem(12, 400, 17)
    want_kf = util.keyframe_levels(oracle, b)
    want_ba = util.run_ba(oracle, prob)
    pc = synth.make_pose_case(n=500)
    want_pose = host.Context(lib=oracle).pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"])[0]
    errors = []

    def tracker():
        try:
            ctx = host.Context(lib=hip)
            kf = host.KeyFrame(ctx)
            for _ in range(30):
                kf.MakeKeyFrame_Lite(b)
                util.assert_levels_equal([kf.level(l) for l in range(4)], want_kf)
                p = ctx.pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"])[0]
                assert np.allclose(p, want_pose, atol=1e-10)
        except Exception as e:  # noqa: BLE001
            errors.append(("tracker", e))

    def mapmaker():
        try:
            for _ in range(6):
                util.assert_ba_equal(util.run_ba(hip, prob), want_ba)
        except Exception as e:  # noqa: BLE001
            errors.append(("mapmaker", e))

    ts = [threading.Thread(target=tracker), threading.Thread(target=mapmaker)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


@pytest.mark.parametrize("case", ["synthetic", "noise", "odd"])
def test_keyframe_rest_matches_oracle(hip, oracle, case):
    rng = np.random.default_rng(13)
    im = {"synthetic": synth.make_frame(), "noise": rng.integers(0, 256, (480, 640), dtype=np.uint8),
          "odd": rng.integers(0, 256, (77, 101), dtype=np.uint8)}[case]
    out = []
    for lib in (hip, oracle):
        ctx = host.Context(lib=lib, size=(im.shape[1], im.shape[0]))
        kf = host.KeyFrame(ctx).MakeKeyFrame_Lite(im)
        out.append(kf.MakeKeyFrame_Rest(min_shi_tomasi=70.0))
        if lib is hip:                                   # second call on the same handle: state is reset
            kf.MakeKeyFrame_Lite(im)
            again = kf.MakeKeyFrame_Rest()
            for l in range(4):
                assert np.array_equal(again[l]["max_corners"], out[0][l]["max_corners"])
    for l in range(4):
        assert np.array_equal(out[0][l]["max_corners"], out[1][l]["max_corners"]), l       # bit-exact, raster order
        assert np.allclose(out[0][l]["st_scores"], out[1][l]["st_scores"], rtol=1e-12, atol=1e-12)
        assert np.array_equal(out[0][l]["candidates"], out[1][l]["candidates"])
    assert len(out[0][0]["candidates"]) > 0


def test_refind_common_matches_oracle(hip, oracle):
    """MapMaker::ReFind_Common (src/MapMaker.cc:943-1020) batched over the map points of one keyframe: 1244 points of the
    TrackMap scenario (junk points behind the camera / outside the view / with degenerate pixel vectors included) searched
    in frame B at its true pose and at a pose a few pixels off (range 4 is tight: many then fail and go to never-retry)"""
    from tests import golden_util as G
    out = {}
    for name, lib in (("hip", hip), ("oracle", oracle)):
        ctx = host.Context(lib=lib)
        a, b = synth.make_frame_pair()
        kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
        kfb = host.KeyFrame(ctx).MakeKeyFrame_Lite(b)
        case = synth.make_trackmap_case([kfa.level(l) for l in range(4)])
        pf = host.PatchFinder(ctx)
        out[name] = [pf.ReFind(kfb, pose, case["world"], case["pixel_right_w"], case["pixel_down_w"], kfa, case["src_level"], case["center"])
                     for pose in (case["cur_pose"], case["pose_in"])]
    for rh, ro in zip(out["hip"], out["oracle"]):
        G.assert_refind_equal(rh, ro["found"], ro["level"], ro["sub_pix"], ro["never_retry"], ro["root_pos"])
    assert out["oracle"][0]["found"].sum() > 1000 and out["oracle"][1]["found"].sum() < out["oracle"][0]["found"].sum()


def test_refind_pairs_through_one_patchfinder_matches_oracle(hip, oracle):
    """MapMaker::ReFind_Common in the reference's call patterns (src/MapMaker.cc:1046-1082; VERDICT r2 missing 3) through ONE
    PatchFinder (:977) whose state outlives the calls: (1) ReFindNewlyMade — 60 new points, each against twelve keyframes in
    turn (frame B at poses that are mostly a fraction of a millimetre apart, so the finder keeps its template, plus one
    pushed towards the scene and one far off); (2) ReFindFromFailureQueue — (keyframe, point) pairs sorted by keyframe, a
    different point every time: every template re-made; (3) the last point again against two keyframes: continues the
    previous call's state."""
    from tests import golden_util as G
    res = {}
    for name, lib in (("hip", hip), ("oracle", oracle)):
        ctx = host.Context(lib=lib)
        a, b = synth.make_frame_pair()
        kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
        kfb = host.KeyFrame(ctx).MakeKeyFrame_Lite(b)
        case = synth.make_trackmap_case([kfa.level(l) for l in range(4)])
        base = np.array(case["cur_pose"], dtype=np.float64)
        z = float(np.median(case["world"] @ base[6:9] + base[11]))

        def moved(dz=0.0, dx=0.0):
            q = base.copy()
            q[9] += dx
            q[11] += dz
            return q
        poses = [base, moved(dz=2e-4 * z), moved(dx=1e-4 * z), moved(dz=-0.25 * z), moved(dz=2e-4 * z), moved(dz=4e-4 * z), case["pose_in"],
                 moved(dx=-1e-4 * z), base, moved(dz=1e-4 * z), moved(dz=0.4 * z), moved(dz=0.4002 * z)]
        rng = np.random.default_rng(77)
        pts = rng.choice(len(case["world"]), size=60, replace=False)
        idx = np.repeat(pts, len(poses))
        pp = np.tile(np.array(poses), (len(pts), 1))
        args = lambda ix: (case["world"][ix], case["pixel_right_w"][ix], case["pixel_down_w"][ix], kfa, case["src_level"][ix], case["center"][ix], ix)
        rf = host.ReFinder(ctx)
        out = [rf.find(host.ReFinder.pairs([kfb] * len(idx), pp, *args(idx), skip=(rng.random(len(idx)) < 0.05).astype(np.int32)))]
        # failure queue: sorted by keyframe, then point
        q_pts = np.sort(rng.choice(len(case["world"]), size=150, replace=False))
        q_kfs = [kfa] * 70 + [kfb] * 80
        q_pose = np.array([case["pose_in"]] * 70 + [base] * 80)
        out.append(rf.find(host.ReFinder.pairs(q_kfs, q_pose, *args(q_pts))))
        last = q_pts[-1:]
        out.append(rf.find(host.ReFinder.pairs([kfb, kfb], np.array([base, moved(dz=1e-4 * z)]), *args(np.repeat(last, 2)))))
        res[name] = out
    for (rh, kh), (ro, ko) in zip(res["hip"], res["oracle"]):
        assert np.array_equal(kh, ko)
        G.assert_refind_equal(rh, ro["found"], ro["level"], ro["sub_pix"], ro["never_retry"], ro["root_pos"])
    k0 = res["oracle"][0][1]
    assert 0.4 * len(k0) < k0.sum() < 0.9 * len(k0)             # most consecutive keyframes keep the template
    assert res["oracle"][1][1].sum() == 0                        # a different point every time: never kept
    assert list(res["oracle"][2][1]) == [1, 1]                   # the state came along from the previous call
    assert res["oracle"][0][0]["found"].sum() > 200


@pytest.mark.parametrize("cams,pts,lo,hi", [(10, 400, 20, 120), (6, 300, 5, 290), (8, 500, 3, 70), (12, 1000, 100, 101)],
                         ids=lambda v: str(v))
def test_bundle_points_without_measurements(hip, oracle, cams, pts, lo, hi):
    """long stretches of points that nobody measures (never measured, or purged earlier) between observed ones: the device
    numbers the observed points densely — the kernels fetch a chunk's points as one run of consecutive ids, which such a gap
    used to break (wrong coordinates, silently) — and an unobserved point keeps its position (src/Bundle.cc:341-359)"""
    p = synth.make_ba_problem(cams, pts, 21)
    keep = ~((p["pt_idx"] >= lo) & (p["pt_idx"] < hi))
    q = {k: (v[keep] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in p.items()}
    a, b = util.run_ba(hip, q, max_iterations=8), util.run_ba(oracle, q, max_iterations=8)
    util.assert_ba_equal(a, b, rel=1e-6)
    assert a["accepted"] > 0
    assert np.array_equal(a["points"][lo:hi], q["points"][lo:hi])          # untouched
    assert not np.array_equal(a["points"][hi:hi + 5], q["points"][hi:hi + 5])   # the others moved


def test_persistent_solve_under_contention(hip):
    """The persistent camera solve's workgroups wait for each other through flags (csrc/ldlt_chain.inc): with other queues
    keeping the device busy every one of them must still be dispatched — a spin that gives up surfaces as PTAM_E_HIP — and
    in deterministic mode the results must be bit-identical to the quiet runs (one chain: 50 dense cameras; two chains and a
    middle part: 140 cameras, window 12).  tests/tools/stress_chain_contention.py is the long form."""
    import threading
    import time
    probs = [synth.make_ba_problem(n_cams=50, n_pts=1500, seed=7), synth.make_ba_problem(n_cams=140, n_pts=2500, seed=8, window=12)]
    quiet = [util.run_ba(hip, p, max_iterations=5, deterministic=1) for p in probs]
    stop = []

    def load():
        ctx = host.Context(lib=hip)
        a, b = synth.make_frame_pair()
        kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
        case = synth.make_trackmap_case([kfa.level(l) for l in range(4)])
        tr = host.Tracker(ctx, len(case["world"]))
        tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], kfa, case["src_level"], case["center"])
        kfb, d_im, opts = host.KeyFrame(ctx), host.DevBuf(ctx, b), tr.opts()
        while not stop:
            tr.set_shuffle(case["shuffle_levels"], case["shuffle_fine"])
            tr.TrackFrame(kfb, d_im, case["pose_in"], opts)

    threads = [threading.Thread(target=load) for _ in range(4)]
    for t in threads:
        t.start()
    try:
        t0, runs = time.time(), 0
        while time.time() - t0 < 3.0:
            for p, q in zip(probs, quiet):
                r = util.run_ba(hip, p, max_iterations=5, deterministic=1)
                assert len(r["trials"]) == len(q["trials"])
                assert np.array_equal(r["poses"], q["poses"]) and np.array_equal(r["points"], q["points"])
                runs += 1
        assert runs >= 4
    finally:
        stop.append(1)
        for t in threads:
            t.join()



@pytest.mark.parametrize("case", ["chain_64x900", "chain_100x900_w40", "twisted_128x1200_w24_chains", "twisted_181x2500_w12_f3"])
def test_persistent_solve_that_gives_up_is_repeated_per_column(hip, oracle, case):
    """VERDICT r3 item 6 / ADVICE r3: a wait of the persistent camera solve that gives up (csrc/ldlt_chain.inc) must not end the
    adjustment — the mapmaker would reset the map.  With a spin limit of ONE look (PTAM_CH_SPIN_LIMIT, read once per process:
    hence the subprocess) every persistent solve form — one chain, banded one chain, two chains + middle — is void at its first
    hand-off; the trial is then run again with the launch-per-block-column form and the adjustment goes on: same trials, same
    state as the oracle, the fallback counted once."""
    r = util.run_ba_subprocess(BA_CASES[case], env={"PTAM_CH_SPIN_LIMIT": "1"})
    ro = util.run_ba(oracle, synth.make_ba_problem(**BA_CASES[case]))
    util.assert_ba_equal(r, ro, rel=1e-6)
    assert r["solve_fallbacks"] == 1 and r["accepted"] > 0
    quiet = util.run_ba(hip, synth.make_ba_problem(**BA_CASES[case]))
    assert quiet["solve_fallbacks"] == 0


def test_banded_system_of_more_than_a_thousand_cameras(hip):
    """the right-hand-side workgroup of the persistent solve indexed its LDS vector by ABSOLUTE row; in the
    middle launch of a two-ended elimination that ran past the buffer once b_start > 33 (band + 1) — ~1040 cameras at band 2 —
    and the last rows' solution was silently wrong.  1150 cameras, 6-camera window (216 block rows, band 2): the persistent
    forms must agree with the launch-per-block-column forms (PTAM_LDLT_NO_CHAIN, another process) trial by trial.  (The oracle's
    dense 6894^3 / 3 factorisation would take minutes per trial: the per-column form is what the oracle pins, at 640 cameras.)"""
    case = dict(n_cams=1150, n_pts=6000, seed=77, window=6)
    a = util.run_ba(hip, synth.make_ba_problem(**case), max_iterations=4)
    b = util.run_ba_subprocess(case, env={"PTAM_LDLT_NO_CHAIN": "1"}, opts=dict(max_iterations=4))
    util.assert_ba_equal(a, b, rel=1e-9, abs_state=1e-9)
    t = a["trials"]
    assert len(t) == 4 and a["accepted"] > 0 and np.isfinite(t["err_new"]).all()
    assert all(x["err_new"] < x["err_old"] for x in t if x["accepted"])


def test_two_bundles_with_persistent_solves_side_by_side(hip):
    """the persistent solve needs all of its workgroups resident; two bundles of 20+ block rows adjusting at
    the same moment on one device (two mapmaker-like threads, a context each) could each be granted part of theirs.  Only one
    bundle at a time takes the persistent form (the other one the launch-per-block-column form for that call), every bundle has an
    XCD of its own: both finish, neither falls back, results as when run alone."""
    import threading
    probs = [synth.make_ba_problem(n_cams=100, n_pts=900, seed=44, window=40), synth.make_ba_problem(n_cams=128, n_pts=1200, seed=45, window=24)]
    alone = [util.run_ba(hip, p, max_iterations=6) for p in probs]
    got = {}

    def work(i):
        got[i] = [util.run_ba(hip, probs[i], max_iterations=6) for _ in range(6)]

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert all(not t.is_alive() for t in th)
    for i in range(2):
        assert len(got[i]) == 6
        for r in got[i]:
            util.assert_ba_equal(r, alone[i], rel=1e-8, abs_state=1e-8)
            assert r["solve_fallbacks"] == 0


# the fuzzers' off-tolerance cases before a referee.  On these problems — short or thin camera chains held by
# one fixed camera, triplicated points, 15 % outliers, 5 % point noise — product and oracle take the same discrete trajectory
# but their robust errors drift apart by 1e-5 .. 3e-4 in the late trials.  The referee is the oracle's Bundle
# in x87 extended precision (oracle/referee.cc, eps 5e-20): against it BOTH are off by the same order — the distance is the
# problem's conditioning, amplified last-bit differences of early trials, not an error of either side.  (The three worst of
# tests/tools/referee_fuzz.py: seeds 101 / 240 cases and 11 / 150 --big.)
REFEREE_CASES = {
    "thin_27x633_w2_dup3": (dict(n_cams=27, n_pts=633, seed=5027, window=2, n_fixed=1, outlier_frac=0.15, pt_noise=0.002, dup=3), _abi.EST_TUKEY),
    "thin_35x681_w4_f3": (dict(n_cams=35, n_pts=681, seed=5060, window=4, n_fixed=3, outlier_frac=0.15, pt_noise=0.05, dup=1), _abi.EST_TUKEY),
    "chain_91x509_w12_dup3": (dict(n_cams=91, n_pts=509, seed=5136, window=12, n_fixed=1, outlier_frac=0.15, pt_noise=0.05, dup=3), _abi.EST_CAUCHY),
}


def _trial_distance(a, b, n):
    w = 0.0
    for x, y in zip(a[:n], b[:n]):
        for k in ("sigma_sq", "err_old", "err_new"):
            w = max(w, abs(x[k] - y[k]) / max(abs(y[k]), 1e-300))
    return w


@pytest.mark.parametrize("name", list(REFEREE_CASES))
def test_ill_conditioned_bundles_before_the_extended_precision_referee(hip, oracle, name):
    from tests import referee_lib
    case, est = REFEREE_CASES[name]
    prob = synth.make_ba_problem(**case)
    ref = referee_lib.run_ba(prob, estimator=est)
    orc = util.run_ba(oracle, prob, estimator=est)
    n = len(ref["trials"])
    assert len(orc["trials"]) == n
    d_orc = _trial_distance(orc["trials"], ref["trials"], n)
    for det in (1, 0):       # the fixed-order mode (one trajectory per build) and the default one (camera sums by LDS atomics)
        got = util.run_ba(hip, prob, estimator=est, deterministic=det)
        assert len(got["trials"]) == n
        for x, y in zip(got["trials"], ref["trials"]):          # the discrete trajectory is the referee's
            assert abs(x["lambda"] - y["lambda"]) <= 1e-12 * abs(y["lambda"]) and x["accepted"] == y["accepted"] and x["n_bad"] == y["n_bad"]
        d_got = _trial_distance(got["trials"], ref["trials"], n)
        print(f"{name} deterministic={det}: product <-> referee {d_got:.2e}, oracle <-> referee {d_orc:.2e}")
        # both are off by the same order; the product is not the outlier of the three
        assert d_got <= 10 * max(d_orc, 1e-9)
        assert np.array_equal(got["outliers"], ref["outliers"])


# a seeded slice of the fuzzers (tests/tools/fuzz_ba.py, referee_fuzz.py: the same generator) inside the gated
# suite.  The rule, per case: the product is within the 1e-6 the path promises of the checker, trial by trial — or, where the two
# differ by more (2 % of random shapes: thin, ill-conditioned chains on which last-bit differences of an early trial are amplified),
# the x87 extended-precision referee arbitrates: over the trials all three walk together the product must be no farther from the
# referee than the checker is (factor 10 as in the named cases above), and it must follow the referee's discrete trajectory at
# least as long as the checker does.
def _fuzz_cases(big, seed, n):
    rng = np.random.default_rng(seed)
    for i in range(n):
        n_cams = int(rng.integers(60, 220)) if big else int(rng.integers(2, 90))
        n_pts = int(rng.integers(200, 700)) if big else int(rng.integers(3, 900))
        window = (int(rng.integers(4, 60)) if big else (None if rng.random() < 0.35 else int(rng.integers(2, max(3, n_cams)))))
        case = dict(n_cams=n_cams, n_pts=n_pts, seed=5000 + i, window=window, n_fixed=int(rng.integers(1, min(4, n_cams))),
                    outlier_frac=float(rng.choice([0.0, 0.02, 0.15])), pt_noise=float(rng.choice([0.002, 0.01, 0.05])),
                    dup=int(rng.choice([1, 1, 1, 3])))
        est = [_abi.EST_TUKEY, _abi.EST_CAUCHY, _abi.EST_HUBER][i % 3]
        mi = int(rng.choice([20, 20, 3, 7]))
        yield i, case, est, mi


def _common_prefix(a, b):
    n = 0
    for x, y in zip(a, b):
        if not (abs(x["lambda"] - y["lambda"]) <= 1e-12 * abs(y["lambda"]) and x["accepted"] == y["accepted"] and x["n_bad"] == y["n_bad"]):
            break
        n += 1
    return n


@pytest.mark.parametrize("kind,seed,n", [("small", 101, 60), ("big", 11, 20)])
def test_fuzz_slice_within_tolerance_or_as_close_to_the_referee_as_the_checker(hip, oracle, kind, seed, n):
    from tests import referee_lib
    refereed = []
    for i, case, est, mi in _fuzz_cases(kind == "big", seed, n):
        prob = synth.make_ba_problem(**case)
        if len(prob["cam_idx"]) == 0:
            continue
        # (fixed-order camera sums: the product's trajectory on these ill-conditioned shapes is then the same in every run of the
        #  suite — with the default LDS atomics a case at the noise floor may leave the referee's trajectory a trial earlier in one
        #  run than in the next)
        a = util.run_ba(hip, prob, estimator=est, max_iterations=mi, deterministic=1)
        b = util.run_ba(oracle, prob, estimator=est, max_iterations=mi)
        try:
            util.assert_ba_equal(a, b, rel=1e-6)
            continue
        except AssertionError:
            pass
        r = referee_lib.run_ba(prob, estimator=est, max_iterations=mi)
        pa, pb = _common_prefix(a["trials"], r["trials"]), _common_prefix(b["trials"], r["trials"])
        n3 = min(pa, pb)
        dp, do = _trial_distance(a["trials"], r["trials"], n3), _trial_distance(b["trials"], r["trials"], n3)
        refereed.append((i, n3, dp, do))
        assert pa >= pb, (i, case, pa, pb)                      # leaves the referee's trajectory no earlier than the checker
        assert dp <= 10 * max(do, 1e-9), (i, case, dp, do)      # and is no farther from it
    print(f"{kind}: {len(refereed)} of {n} cases before the referee: {refereed}")
    assert len(refereed) <= max(3, n // 8)                      # (2 % expected: more would be a regression of the path, not of its conditioning)


@pytest.mark.parametrize("shape", [(50, 5000, None), (20, 3000, None), (200, 26000, 16), (9, 400, None)])
def test_schur_work_lists_fit_the_resident_slots(shape):
    """The Schur tile kernel's work split (ba_prepare_impl: cost model fitted to stamps, budget by bisection) must never make
    more workgroups than the chip holds at once — 2 per CU, 512 — or a second round of them doubles the launch; and a
    workgroup should stay at a few segments (each costs ~5 us of pipeline fill and partial-tile traffic).  Read from the
    library's own PTAM_DEBUG_SCHUR report, in a process of its own (the variable is looked at once)."""
    import os
    import re
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from ptam_cg_amd import host, synth\n"
            "from ptam_cg_amd._lib import load\n"
            "ctx = host.Context(lib=load())\n"
            "ba = synth.load_into(host.Bundle(ctx), synth.make_ba_problem(%d, %d, 11, window=%r))\n"
            "ba.prepare(); ctx.sync(); ba.close()\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), shape[0], shape[1], shape[2])
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PTAM_DEBUG_SCHUR="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"schur: (\d+) segments, (\d+) workgroups", r.stderr)
    assert m, r.stderr[-2000:]
    segs, wgs = int(m.group(1)), int(m.group(2))
    assert 0 < wgs <= 512
    assert segs <= 2.5 * wgs
