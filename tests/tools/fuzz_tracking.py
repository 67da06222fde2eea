import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import faulthandler; faulthandler.dump_traceback_later(150, exit=True)
import numpy as np, torch
from ptam_cg_amd import _abi, host, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
hip, oracle = load(), load_oracle()
rng = np.random.default_rng(11)
a, b = synth.make_frame_pair()
res = {}
for name, lib in (("hip", hip), ("oracle", oracle)):
    ctx = host.Context(lib=lib)
    kfa, kfb = host.KeyFrame(ctx).MakeKeyFrame_Lite(a), host.KeyFrame(ctx).MakeKeyFrame_Lite(b)
    pf = host.PatchFinder(ctx)
    r = np.random.default_rng(11)
    n = 4000
    q = np.zeros(n, dtype=host.PATCH_QUERY_DT)
    q["x"] = r.integers(-50, 700, n); q["y"] = r.integers(-50, 540, n); q["level"] = r.integers(-1, 4, n); q["range"] = r.integers(0, 60, n)
    t = r.integers(0, 256, (n, 64)).astype(np.uint8)
    # half of the templates: real windows of frame A so that matches exist
    lv0 = kfa.level(0)["im"]
    for i in range(0, n, 2):
        x, y = r.integers(4, 636), r.integers(4, 476)
        t[i] = lv0[y - 4:y + 4, x - 4:x + 4].reshape(64)
        q["x"][i], q["y"][i], q["level"][i] = x + 3, y - 2, 0
    fp = pf.FindPatchCoarse(kfb, q, t)
    pos = np.stack([r.uniform(-5, 645, n), r.uniform(-5, 485, n)], axis=1)
    sp = pf.SubPix(kfb, pos, r.integers(-1, 4, n), t, 8)
    tc = synth.make_template_cases((640, 480), n=3000, seed=99)
    tc["warp_inverse"] *= r.uniform(0.3, 3.0, (3000, 1))
    tm, tr = pf.MakeTemplateCoarseCont(kfa, tc["src_level"], tc["center"], tc["search_level"], tc["warp_inverse"])
    res[name] = (fp, sp, tm, tr)
(fh, sh, th, rh), (fo, so, to, ro) = res["hip"], res["oracle"]
for f in ("found", "best_ssd", "best_x", "best_y", "n_scored"):
    print("patch", f, np.array_equal(fh[f], fo[f]))
print("patch pos", np.array_equal(fh["pos"], fo["pos"]), "found", int(fo["found"].sum()))
print("subpix conv", np.array_equal(sh["converged"], so["converged"]), "its", np.array_equal(sh["iterations"], so["iterations"]),
      "pos", float(np.nanmax(np.abs(sh["pos"] - so["pos"]))), "converged", int(so["converged"].sum()))
print("templates", np.array_equal(th, to), [bool(np.array_equal(rh[f], ro[f])) for f in ("bad", "n_outside", "sum", "sum_sq", "m2")], "outside", int((ro["n_outside"] > 0).sum()))
d = np.abs(sh["pos"] - so["pos"]).max(axis=1)
c = so["converged"] != 0
print("subpix diff converged max", float(np.nanmax(d[c])) if c.any() else None, "non-converged max", float(np.nanmax(d[~c])), "count>1e-9 (conv)", int((d[c] > 1e-9).sum()), "(non)", int((d[~c] > 1e-9).sum()))
i = int(np.nanargmax(d)); print("worst", i, sh[i], so[i])
