import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import faulthandler; faulthandler.dump_traceback_later(100, exit=True)
import numpy as np, torch
from ptam_cg_amd import _abi, host, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
hip, oracle = load(), load_oracle()
rng = np.random.default_rng(5)
a, _ = synth.make_frame_pair()
def kf_case(w, h):
    im = np.ascontiguousarray(a[:h, :w])
    out = []
    for lib in (hip, oracle):
        ctx = host.Context(lib=lib, size=(w, h)); kf = host.KeyFrame(ctx).MakeKeyFrame_Lite(im)
        lv = [kf.level(l) for l in range(4)]
        rest = None
        out.append(lv); kf.close(); ctx.close()
    ok = all(np.array_equal(x["im"], y["im"]) and np.array_equal(x["corners"], y["corners"]) and np.array_equal(x["rowlut"], y["rowlut"]) for x, y in zip(*out))
    print("KF", w, h, "OK" if ok else "DIFF", [len(x["corners"]) for x in out[0]], flush=True)
for w, h in ((640, 480), (322, 246), (100, 75), (64, 64), (129, 67), (72, 56), (640, 482), (48, 40)):
    try: kf_case(w, h)
    except Exception as e: print("KF", w, h, "EXC", type(e).__name__, str(e)[:120], flush=True)
pc = synth.make_pose_case(n=3000)
for n in (1, 2, 7, 64, 255, 1024, 1025, 3000):
    try:
        ph, fh, uh = host.Context(lib=hip).pose_gn(pc["world"][:n], pc["found"][:n], pc["sqrt_inv_noise"][:n], pc["init_pose"])
        po, fo, uo = host.Context(lib=oracle).pose_gn(pc["world"][:n], pc["found"][:n], pc["sqrt_inv_noise"][:n], pc["init_pose"])
        ok = np.allclose(ph, po, rtol=0, atol=1e-9, equal_nan=True) and np.array_equal(fh, fo)
        print("POSE n", n, "OK" if ok else "DIFF", float(np.nanmax(np.abs(ph - po))), flush=True)
    except Exception as e: print("POSE n", n, "EXC", type(e).__name__, str(e)[:120], flush=True)
pv = synth.make_pvs_case(n=300)
for n in (0, 1, 65):
    try:
        rh, ch = host.Context(lib=hip).track_pvs(pv["world"][:n], pv["pixel_right_w"][:n], pv["pixel_down_w"][:n], pv["pose"])
        ro, co = host.Context(lib=oracle).track_pvs(pv["world"][:n], pv["pixel_right_w"][:n], pv["pixel_down_w"][:n], pv["pose"])
        print("PVS n", n, "OK" if np.array_equal(rh["level"], ro["level"]) and np.array_equal(ch, co) else "DIFF", flush=True)
    except Exception as e: print("PVS n", n, "EXC", type(e).__name__, str(e)[:120], flush=True)
