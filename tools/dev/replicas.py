"""Aggregate tracked frames/s of k independent trackers on one device, driven natively (ptam_bench_track_frames).
usage: [GPU_MAX_HW_QUEUES=n] python tools/dev/replicas.py [k ...]"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load()
ks = [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32]
a, b = synth.make_frame_pair()
ctx0 = host.Context(lib=hip)
kfa0 = host.KeyFrame(ctx0).MakeKeyFrame_Lite(a)
case = synth.make_trackmap_case([kfa0.level(l) for l in range(4)])
sl = np.ascontiguousarray(case["shuffle_levels"], dtype=np.int32); sf = np.ascontiguousarray(case["shuffle_fine"], dtype=np.int32)
pose = np.ascontiguousarray(case["pose_in"]); raw = lambda h: h.value if hasattr(h, "value") else int(h)
for k in ks:
    ws = []
    for _ in range(k):
        cx = host.Context(lib=hip); ka = host.KeyFrame(cx).MakeKeyFrame_Lite(a); di = host.DevBuf(cx, b)
        tr = host.Tracker(cx, len(case["world"]))
        tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], ka, case["src_level"], case["center"])
        ws.append((cx, ka, host.KeyFrame(cx), tr, di))
    opts = ws[0][3].opts()
    trs = (C.c_void_p * k)(*[raw(w[3].h) for w in ws]); kfs = (C.c_void_p * k)(*[raw(w[2].h) for w in ws]); dis = (C.c_void_p * k)(*[raw(w[4].p) for w in ws])
    secs = C.c_double(); nf = int(os.environ.get("NF", max(40, 2000 // k)))
    for rep in range(2):
        ctx0._check(hip.bench_track_frames(k, trs, kfs, dis, pose.ctypes.data_as(C.POINTER(C.c_double)), opts.ctypes.data_as(C.c_void_p),
                                           sl.ctypes.data_as(C.c_void_p), sf.ctypes.data_as(C.c_void_p), nf, C.byref(secs)), "bench")
    print(f"contexts {k:3d}: {k * nf / secs.value:9.0f} frames/s  ({secs.value / nf * 1e6:.0f} us per frame per context)", flush=True)
    if os.environ.get("BATCH"):
        rounds = max(20, 2000 // k)
        for groups in [int(g) for g in os.environ.get("NGROUPS", "1").split(",")]:
            if groups > k: continue
            for rep in range(2):
                ctx0._check(hip.bench_track_batch(k, trs, kfs, dis, pose.ctypes.data_as(C.POINTER(C.c_double)), opts.ctypes.data_as(C.c_void_p),
                                                  sl.ctypes.data_as(C.c_void_p), sf.ctypes.data_as(C.c_void_p), rounds, groups, C.byref(secs)), "bench_batch")
            print(f"   batched {k:3d} in {groups} group(s): {k * rounds / secs.value:9.0f} frames/s  ({secs.value / rounds * 1e6:.0f} us per round)", flush=True)
