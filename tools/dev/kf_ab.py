"""stand-alone MakeKeyFrame_Lite of a device-resident frame, microseconds per call (median of 7 blocks of 300)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load(); ctx = host.Context(lib=hip)
a, b = synth.make_frame_pair()
kf = host.KeyFrame(ctx); d = host.DevBuf(ctx, b)
for _ in range(500): ctx._check(hip.make_keyframe_lite_dev(ctx.h, kf.h, d.p), "kf")
ctx.sync(); bl = []
for _ in range(7):
    t0 = time.perf_counter()
    for _ in range(300): ctx._check(hip.make_keyframe_lite_dev(ctx.h, kf.h, d.p), "kf")
    ctx.sync(); bl.append((time.perf_counter() - t0) / 300 * 1e6)
bl.sort(); print("keyframe us: median %.2f  min %.2f  max %.2f" % (bl[3], bl[0], bl[-1]))
