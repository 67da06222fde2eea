import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from ptam_cg_amd import _abi, host, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util
hip, oracle = load(), load_oracle()
cases = [
 (dict(n_cams=11, n_pts=593, seed=5021, window=2, n_fixed=2, outlier_frac=0.02, pt_noise=0.05, dup=1), 0, 20),
 (dict(n_cams=3, n_pts=650, seed=5065, window=2, n_fixed=1, outlier_frac=0.15, pt_noise=0.002, dup=3), 2, 20),
 (dict(n_cams=12, n_pts=569, seed=5091, window=9, n_fixed=1, outlier_frac=0.15, pt_noise=0.01, dup=1), 1, 20),
 (dict(n_cams=4, n_pts=597, seed=5120, window=2, n_fixed=3, outlier_frac=0.15, pt_noise=0.002, dup=1), 0, 7),
 (dict(n_cams=2, n_pts=85, seed=5128, window=None, n_fixed=1, outlier_frac=0.15, pt_noise=0.002, dup=1), 2, 20),
 (dict(n_cams=72, n_pts=315, seed=5141, window=6, n_fixed=2, outlier_frac=0.15, pt_noise=0.05, dup=3), 0, 20),
]
for case, est, mi in cases:
    prob = synth.make_ba_problem(**case)
    a = util.run_ba(hip, prob, estimator=est, max_iterations=mi); b = util.run_ba(oracle, prob, estimator=est, max_iterations=mi)
    print("CASE", case["n_cams"], case["n_pts"], "w", case["window"], "est", est, "M", len(prob["cam_idx"]), "trials", len(a["trials"]), len(b["trials"]),
          "acc", a["accepted"], b["accepted"], "conv", a["converged"], b["converged"], "outl", len(a["outliers"]), len(b["outliers"]))
    for i, (x, y) in enumerate(zip(a["trials"], b["trials"])):
        rel = lambda k: abs(x[k] - y[k]) / max(abs(y[k]), 1e-300)
        flag = "" if (x["lambda"] == y["lambda"] and x["accepted"] == y["accepted"] and x["n_bad"] == y["n_bad"]) else "  <<<"
        print(f"  {i:2d} lam {x['lambda']:.3e}/{y['lambda']:.3e} acc {x['accepted']}/{y['accepted']} nbad {x['n_bad']}/{y['n_bad']} rel sig {rel('sigma_sq'):.1e} old {rel('err_old'):.1e} new {rel('err_new'):.1e}{flag}")
    print("  pose maxdiff", np.nanmax(np.abs(a["poses"] - b["poses"])), "pts maxdiff", np.nanmax(np.abs(a["points"] - b["points"])),
          "outliers equal", np.array_equal(a["outliers"], b["outliers"]))
