"""a batch whose measurement lists exceed the register-resident pose kernel (general kernel in the batch): against single calls"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load()
a, b = synth.make_frame_pair()
ctx0 = host.Context(lib=hip); kfa0 = host.KeyFrame(ctx0).MakeKeyFrame_Lite(a)
ws = []
for i in range(3):
    case = synth.make_trackmap_case([kfa0.level(l) for l in range(4)], counts=(1000, 900, 700, 500), seed=300 + i)
    cx = host.Context(lib=hip); ka = host.KeyFrame(cx).MakeKeyFrame_Lite(a)
    tr = host.Tracker(cx, len(case["world"]))
    tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], ka, case["src_level"], case["center"])
    ws.append((cx, ka, host.KeyFrame(cx), tr, host.DevBuf(cx, b), case))
opts = ws[0][3].opts(max_patches=2000)
single = []
for cx, ka, kb, tr, di, case in ws:
    tr.set_shuffle(case["shuffle_levels"], case["shuffle_fine"]); single.append(tr.TrackFrame(kb, di, case["pose_in"], opts).copy())
for cx, ka, kb, tr, di, case in ws:
    # (the same history on both sides: set_map starts every point's PatchFinder afresh — the trackers keep them since round 3)
    tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], ka, case["src_level"], case["center"])
    tr.set_shuffle(case["shuffle_levels"], case["shuffle_fine"])
res = host.Tracker.TrackFramesBatch([w[3] for w in ws], [w[2] for w in ws], [w[4] for w in ws], [w[5]["pose_in"] for w in ws], opts)
for i in range(3):
    # (the batch's pose kernels take their sums in another order than the single call's: everything but the pose and the depth
    #  sums to the bit, those to rounding)
    same = all(np.allclose(res[i][f], single[i][f], rtol=1e-12, atol=1e-12) if f in ("pose", "depth_sum", "depth_sum_sq") else np.array_equal(res[i][f], single[i][f])
               for f in res.dtype.names)
    print("frame", i, "n_meas", int(res[i]["n_meas"]), "equal to the single call:", same)
    if not same:
        for f in res.dtype.names:
            if not np.array_equal(res[i][f], single[i][f]):
                a_, b_ = np.asarray(res[i][f], dtype=float), np.asarray(single[i][f], dtype=float)
                print("   ", f, "max abs difference", float(np.max(np.abs(a_ - b_))))
