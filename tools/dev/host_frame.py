"""Per-frame cost when the frame arrives in HOST memory (ptam_make_keyframe_lite) against the resident variant."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load(); ctx = host.Context(lib=hip)
a, b = synth.make_frame_pair()
kf = host.KeyFrame(ctx)
d_im = host.DevBuf(ctx, b)
def loop(fn, n=300):
    for _ in range(20): fn()
    ctx.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n // 5): fn()
        ctx.sync()
        ts.append((time.perf_counter() - t0) / (n // 5))
    return sorted(ts)
h = loop(lambda: ctx._check(hip.make_keyframe_lite(ctx.h, kf.h, b.ctypes.data, b.shape[1]), "kf"))
d = loop(lambda: ctx._check(hip.make_keyframe_lite_dev(ctx.h, kf.h, d_im.p), "kf"))
print("host frame: median %.1f us (min %.1f max %.1f) | resident frame: median %.1f us" % (h[2] * 1e6, h[0] * 1e6, h[-1] * 1e6, d[2] * 1e6))
