#!/usr/bin/env python3
"""the moving-camera sequence only (for rocprofv3 --kernel-trace --stats, or for A/B runs): tools/dev/track_seq.py [passes]
prints us per frame (native driver), the kept-template share and the per-stage event times"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402
from ptam_cg_amd import host, synth  # noqa: E402
from ptam_cg_amd._lib import load  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prof = len(sys.argv) > 2 and sys.argv[2] == "stages"
hip = load()
ctx = host.Context(lib=hip)
frames, poses, kim, kpose = synth.make_tracking_frames(64)
kf0 = host.KeyFrame(ctx).MakeKeyFrame_Lite(kim)
m = synth.make_sequence_map([kf0.level(l) for l in range(4)], kpose)
tr = host.Tracker(ctx, len(m["world"]))
tr.set_map(m["world"], m["pixel_right_w"], m["pixel_down_w"], kf0, m["src_level"], m["center"])
kf = host.KeyFrame(ctx)
d_frames = [host.DevBuf(ctx, f) for f in frames]
opts = tr.opts()
mm = tr.motion_model(poses[0])
tr.track_sequence_native(kf, d_frames, mm, opts, m["shuffle_levels"], m["shuffle_fine"], passes=2)
best = None
for _ in range(3):
    secs, st = tr.track_sequence_native(kf, d_frames, mm, opts, m["shuffle_levels"], m["shuffle_fine"], passes=passes, poses_true=poses)
    us = 1e6 * secs / st["frames"]
    best = us if best is None else min(best, us)
print(f"moving sequence: {best:.1f} us/frame ({1e6 / best:.0f} fps), kept {100 * st['templates_reused'] / st['searched']:.0f} % of the templates, "
      f"found {st['measurements'] / st['frames']:.0f} of {st['searched'] / st['frames']:.0f}, coarse on {st['frames_did_coarse'] / st['frames']:.2f}, "
      f"max position error {st['max_position_error_m']:.1e} m")
if prof:
    tr.set_profiling(True)
    tr.track_sequence_native(kf, d_frames, mm, opts, m["shuffle_levels"], m["shuffle_fine"], passes=2)
    s = tr.stage_times()
    print("stages (us, with event overhead):", {k: round(v, 1) for k, v in s.items()}, "sum", round(sum(s.values()), 1))
