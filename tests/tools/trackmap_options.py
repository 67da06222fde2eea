"""TrackMap options beyond the committed cases: a doubled coarse stage (after a recovery the caller doubles CoarseMax / CoarseRange,
src/Tracker.cc:505-516), a patch budget above what the register-resident pose kernel holds, no budget at all — the chain against
the composition through the product's own stage calls (bit for bit)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
from tests import trackmap_ref
from tests.test_gpu_trackmap import _check
hip = load()
for counts, okw in (((800, 300, 80, 40), dict(coarse_max=120, coarse_range=60)),
                    ((1000, 900, 700, 500), dict(max_patches=2000)),
                    ((1000, 900, 700, 500), dict(max_patches=1100, coarse_max=100)),
                    ((400, 200, 60, 30), dict(max_patches=0)),
                    ((800, 300, 80, 40), dict(coarse_subpix_its=0))):
    ctx = host.Context(lib=hip)
    a, b = synth.make_frame_pair()
    kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a); kfb = host.KeyFrame(ctx).MakeKeyFrame_Lite(b)
    case = synth.make_trackmap_case([kfa.level(l) for l in range(4)], counts=counts)
    tr = host.Tracker(ctx, len(case["world"]))
    tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], kfa, case["src_level"], case["center"])
    tr.set_shuffle(case["shuffle_levels"], case["shuffle_fine"])
    res = tr.TrackMap(kfb, case["pose_in"], tr.opts(**okw)); it = tr.iteration_set(); tr.close()
    ref = trackmap_ref.track_map(ctx, kfb, kfa, case, case["pose_in"], case["shuffle_levels"], case["shuffle_fine"], **okw)
    try:
        _check(res, it, ref, strict=True); print(counts, okw, "EQUAL: coarse", res["n_coarse"], "fine", res["n_fine"], "meas", res["n_meas"], "did_coarse", res["did_coarse"])
    except AssertionError as e:
        print(counts, okw, "DIFF", repr(e)[:200])
