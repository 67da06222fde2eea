"""a bundle whose point ids have long gaps (points without any measurement between observed ones): HIP vs oracle"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util
hip, oracle = load(), load_oracle()
for (cams, pts, lo, hi) in ((10, 400, 20, 120), (6, 300, 5, 290), (12, 1000, 100, 101), (8, 500, 3, 70)):
    p = synth.make_ba_problem(cams, pts, 21)
    keep = ~((p["pt_idx"] >= lo) & (p["pt_idx"] < hi))
    q = {k: (v[keep] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in p.items()}
    a = util.run_ba(hip, q, max_iterations=8); b = util.run_ba(oracle, q, max_iterations=8)
    try:
        util.assert_ba_equal(a, b, rel=1e-6); print(cams, pts, "gap", lo, hi, "EQUAL", len(a["trials"]), "trials")
    except AssertionError as e:
        print(cams, pts, "gap", lo, hi, "DIFF", str(e)[:160])
