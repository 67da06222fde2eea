import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from ptam_cg_amd import _abi, host, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util
hip, oracle = load(), load_oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    n_cams = int(rng.integers(2, 90)); n_pts = int(rng.integers(3, 900))
    window = None if rng.random() < 0.35 else int(rng.integers(2, max(3, n_cams)))
    case = dict(n_cams=n_cams, n_pts=n_pts, seed=5000 + i, window=window, n_fixed=int(rng.integers(1, min(4, n_cams))),
                outlier_frac=float(rng.choice([0.0, 0.02, 0.15])), pt_noise=float(rng.choice([0.002, 0.01, 0.05])),
                dup=int(rng.choice([1, 1, 1, 3])))
    prob = synth.make_ba_problem(**case)
    if len(prob["cam_idx"]) == 0: continue
    est = [_abi.EST_TUKEY, _abi.EST_CAUCHY, _abi.EST_HUBER][i % 3]
    mi = int(rng.choice([20, 20, 3, 7]))
    try:
        util.assert_ba_equal(util.run_ba(hip, prob, estimator=est, max_iterations=mi), util.run_ba(oracle, prob, estimator=est, max_iterations=mi), rel=1e-6)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", case, est, mi, str(e)[:300])
print("done, mismatches:", bad)
