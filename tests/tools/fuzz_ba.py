"""Random bundle shapes, HIP vs oracle trial by trial.   usage: fuzz_ba.py [seed] [n] [--diag] [--deterministic] [--big]
--big draws up to 220 cameras with covisibility windows (camera systems of up to 42 block rows: one persistent chain, two
chains + middle part, launch-per-block-column forms) and fewer points, so that the oracle's dense solve stays in seconds.
--deterministic runs the HIP side with ptam_ba_opts.deterministic = 1 (camera sums in a fixed order).
--diag prints, for every mismatching case, the per-trial differences (is the discrete trajectory — lambda, accepted,
n_bad — the same and only the floating-point values drift, or does it fork?)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import _abi, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util
hip, oracle = load(), load_oracle()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
diag = "--diag" in sys.argv
det = 1 if "--deterministic" in sys.argv else 0
rng = np.random.default_rng(int(args[0]) if len(args) > 0 else 7)
bad = 0
for i in range(int(args[1]) if len(args) > 1 else 150):
    big = "--big" in sys.argv
    n_cams = int(rng.integers(60, 220)) if big else int(rng.integers(2, 90)); n_pts = int(rng.integers(200, 700)) if big else int(rng.integers(3, 900))
    window = (int(rng.integers(4, 60)) if big else (None if rng.random() < 0.35 else int(rng.integers(2, max(3, n_cams)))))
    case = dict(n_cams=n_cams, n_pts=n_pts, seed=5000 + i, window=window, n_fixed=int(rng.integers(1, min(4, n_cams))),
                outlier_frac=float(rng.choice([0.0, 0.02, 0.15])), pt_noise=float(rng.choice([0.002, 0.01, 0.05])),
                dup=int(rng.choice([1, 1, 1, 3])))
    prob = synth.make_ba_problem(**case)
    if len(prob["cam_idx"]) == 0: continue
    est = [_abi.EST_TUKEY, _abi.EST_CAUCHY, _abi.EST_HUBER][i % 3]
    mi = int(rng.choice([20, 20, 3, 7]))
    a = util.run_ba(hip, prob, estimator=est, max_iterations=mi, deterministic=det)
    b = util.run_ba(oracle, prob, estimator=est, max_iterations=mi)
    try:
        util.assert_ba_equal(a, b, rel=1e-6)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", i, case, est, mi, str(e)[:200])
        if diag:
            print("   M", len(prob["cam_idx"]), "trials", len(a["trials"]), len(b["trials"]), "acc", a["accepted"], b["accepted"],
                  "conv", a["converged"], b["converged"], "outl", len(a["outliers"]), len(b["outliers"]))
            fork = None
            worst = 0.0
            for t, (x, y) in enumerate(zip(a["trials"], b["trials"])):
                same = x["lambda"] == y["lambda"] and x["accepted"] == y["accepted"] and x["n_bad"] == y["n_bad"]
                if not same and fork is None: fork = t
                if fork is None:
                    for k in ("sigma_sq", "err_old", "err_new"):
                        if not (np.isnan(x[k]) and np.isnan(y[k])):
                            worst = max(worst, abs(x[k] - y[k]) / max(abs(y[k]), 1e-300))
            print("   discrete trajectory", "identical" if fork is None else f"forks at trial {fork}", "| worst rel diff before the fork %.2e" % worst,
                  "| pose maxdiff %.2e pts maxdiff %.2e" % (np.nanmax(np.abs(a["poses"] - b["poses"])), np.nanmax(np.abs(a["points"] - b["points"]))),
                  "| outliers equal", np.array_equal(a["outliers"], b["outliers"]))
print("done, mismatches:", bad)
