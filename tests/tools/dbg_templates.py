import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
out = {}
for name, lib in (("hip", load()), ("oracle", load_oracle())):
    ctx = host.Context(lib=lib)
    a, b = synth.make_frame_pair()
    kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
    kfb = host.KeyFrame(ctx).MakeKeyFrame_Lite(b)
    case = synth.make_trackmap_case([kfa.level(l) for l in range(4)], counts=(400, 200, 60, 30))
    pvs, _ = ctx.track_pvs(case["world"], case["pixel_right_w"], case["pixel_down_w"], case["pose_in"])
    ok = np.flatnonzero(pvs["level"] >= 0)
    pf = host.PatchFinder(ctx)
    tm, tres = pf.MakeTemplateCoarseCont(kfa, case["src_level"][ok], case["center"][ok], pvs["level"][ok], pvs["warp_inverse"][ok])
    q = np.zeros(len(ok), dtype=host.PATCH_QUERY_DT)
    q["x"], q["y"] = pvs["proj"]["image"][ok, 0].astype(np.int32), pvs["proj"]["image"][ok, 1].astype(np.int32)
    q["level"], q["range"] = pvs["level"][ok], 10
    res = pf.FindPatchCoarse(kfb, q, tm)
    f = np.flatnonzero(res["found"] == 1)
    sub = pf.SubPix(kfb, res["pos"][f], q["level"][f], tm[f])
    out[name] = dict(pvs=pvs, tm=tm, tres=tres, res=res, sub=sub, f=f, lev=pvs["level"][ok])
h, o = out["hip"], out["oracle"]
print("pvs warp max diff", np.abs(h["pvs"]["warp_inverse"] - o["pvs"]["warp_inverse"]).max(), "image", np.abs(h["pvs"]["proj"]["image"] - o["pvs"]["proj"]["image"]).max())
d = (h["tm"] != o["tm"])
print("templates differing:", d.any(1).sum(), "of", len(d), "pixels", d.sum(), "levels", np.bincount(h["lev"][d.any(1)], minlength=4))
if d.any():
    i = np.flatnonzero(d.any(1))[0]
    print("example", i, "level", h["lev"][i], "diff pixels", np.flatnonzero(d[i]), h["tm"][i][d[i]], o["tm"][i][d[i]], "m2", h["tres"]["m2"][i], o["tres"]["m2"][i])
print("search equal", np.array_equal(h["res"], o["res"]), "found sets equal", np.array_equal(h["f"], o["f"]))
if np.array_equal(h["f"], o["f"]):
    dd = np.abs(h["sub"]["pos"] - o["sub"]["pos"]).max(1)
    print("subpix max diff", dd.max(), "n>1e-9", (dd > 1e-9).sum(), "of which template differs", d.any(1)[h["f"]][dd > 1e-9].sum())
