#!/usr/bin/env python3
"""frames of the resident TrackMap chain only (for rocprofv3 --kernel-trace --stats): tools/dev/trackmap_only.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
from ptam_cg_amd import host, synth  # noqa: E402
from ptam_cg_amd._lib import load  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fused = len(sys.argv) > 2 and sys.argv[2] == "frame"       # ptam_track_map_frame (keyframe + TrackMap in one call)
hip = load()
ctx = host.Context(lib=hip)
a, b = synth.make_frame_pair()
kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
kfb = host.KeyFrame(ctx)
case = synth.make_trackmap_case([kfa.level(l) for l in range(4)])
tr = host.Tracker(ctx, len(case["world"]))
tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], kfa, case["src_level"], case["center"])
d_im = host.DevBuf(ctx, b)
opts = tr.opts()
for it in range(2):
    t0 = time.perf_counter()
    for _ in range(frames):
        tr.set_shuffle(case["shuffle_levels"], case["shuffle_fine"])
        if fused:
            res = tr.TrackFrame(kfb, d_im, case["pose_in"], opts)
        else:
            ctx._check(hip.make_keyframe_lite_dev(ctx.h, kfb.h, d_im.p), "kf")
            res = tr.TrackMap(kfb, case["pose_in"], opts)
    dt = (time.perf_counter() - t0) / frames
print(f"TrackMap chain ({'one call' if fused else 'two calls'}): {dt*1e6:.1f} us/frame, {1/dt:.0f} fps, found {res['n_meas']} of {sum(res['attempted'])}, did_coarse {res['did_coarse']}")
