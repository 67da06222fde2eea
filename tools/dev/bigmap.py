"""frame chain on a map larger than the set-choice kernel's LDS capacity: tools/dev/bigmap.py [c0 c1 c2 c3]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load(); ctx = host.Context(lib=hip)
counts = tuple(int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (1000, 900, 700, 500)
a, b = synth.make_frame_pair()
kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a); kfb = host.KeyFrame(ctx)
case = synth.make_trackmap_case([kfa.level(l) for l in range(4)], counts=counts)
tr = host.Tracker(ctx, len(case["world"]))
tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], kfa, case["src_level"], case["center"])
d = host.DevBuf(ctx, b); opts = tr.opts()
sl = np.ascontiguousarray(case["shuffle_levels"], dtype=np.int32); sf = np.ascontiguousarray(case["shuffle_fine"], dtype=np.int32)
pose = np.ascontiguousarray(case["pose_in"]); raw = lambda h: h.value if hasattr(h, "value") else int(h)
trs = (C.c_void_p * 1)(raw(tr.h)); kfs = (C.c_void_p * 1)(raw(kfb.h)); dis = (C.c_void_p * 1)(raw(d.p)); secs = C.c_double()
for rep in range(2):
    ctx._check(hip.bench_track_frames(1, trs, kfs, dis, pose.ctypes.data_as(C.POINTER(C.c_double)), opts.ctypes.data_as(C.c_void_p),
                                      sl.ctypes.data_as(C.c_void_p), sf.ctypes.data_as(C.c_void_p), 1000, C.byref(secs)), "bench")
tr.set_shuffle(sl, sf); r = tr.TrackFrame(kfb, d, pose, opts)
print(f"map {len(case['world'])} points: {secs.value / 1000 * 1e6:.1f} us per frame; n_pvs {list(r['n_pvs'])} coarse {r['n_coarse']} fine {r['n_fine']} meas {r['n_meas']}")
