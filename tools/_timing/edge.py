import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from ptam_cg_amd import _abi, host, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util
hip, oracle = load(), load_oracle()
def run(name, prob, **kw):
    try:
        a = util.run_ba(hip, prob, **kw); b = util.run_ba(oracle, prob, **kw)
        util.assert_ba_equal(a, b, rel=1e-6)
        print("OK  ", name, "trials", len(a["trials"]), "acc", a["accepted"], "outl", len(a["outliers"]))
    except Exception as e:
        print("FAIL", name, type(e).__name__, str(e)[:200])
p = synth.make_ba_problem(6, 40, 3)
q = dict(p); q["fixed"] = np.ones_like(p["fixed"]); run("all cameras fixed", q)
q = dict(p); q["fixed"] = np.zeros_like(p["fixed"]); run("no camera fixed (gauge free)", q, max_iterations=5)
# one point only
keep = p["pt_idx"] == 0
q = {k: (v[keep] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in p.items()}; q["points"] = p["points"][:1]; q["points_true"] = p["points_true"][:1]
run("single point", q, max_iterations=5)
# single measurement
q2 = {k: (v[:1] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in q.items()}
run("single measurement", q2, max_iterations=3)
# a point with no measurements, a camera with no measurements
keep = (p["pt_idx"] != 5) & (p["cam_idx"] != 3)
q = {k: (v[keep] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in p.items()}
run("unobserved point + camera", q, max_iterations=6)
# measurements only on fixed camera for some points
keep = ~((p["pt_idx"] < 5) & (p["cam_idx"] != 0))
q = {k: (v[keep] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in p.items()}
run("points seen by the fixed camera only", q, max_iterations=6)
# 65 cameras, every point seen by all: > 64 measurements per point -> block K7
run("65 cams dense (block K7)", synth.make_ba_problem(65, 30, 8), max_iterations=4)
run("64 cams dense (wave K7 edge)", synth.make_ba_problem(64, 30, 8), max_iterations=4)
run("huge outliers", synth.make_ba_problem(8, 200, 9, outlier_frac=0.45), max_iterations=8)
