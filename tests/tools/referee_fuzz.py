"""The off-tolerance cases of fuzz_ba.py before a referee.   usage: referee_fuzz.py [seed] [n] [--big]
Same random shapes as fuzz_ba.py (same generator, same seeds).  Every case on which product and oracle differ by more than the
1e-6 the path promises is run a third time through the extended-precision referee (oracle/referee.cc, tests/referee_lib.py);
per case: up to which trial the three discrete trajectories (lambda, accepted, n_bad) coincide, and over that common prefix the
worst relative distance of the robust errors product <-> referee and oracle <-> referee, and of the final states when all three
took the same trajectory to the end."""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import _abi, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util, referee_lib
hip, oracle = load(), load_oracle()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
big = "--big" in sys.argv
rng = np.random.default_rng(int(args[0]) if len(args) > 0 else 7)


def common_prefix(a, b):
    n = 0
    for x, y in zip(a, b):
        # (the referee's lambda is a long double rounded to the log's double: lambda * 0.3 can differ from the double product in the last bit)
        if not (abs(x["lambda"] - y["lambda"]) <= 1e-12 * abs(y["lambda"]) and x["accepted"] == y["accepted"] and x["n_bad"] == y["n_bad"]):
            break
        n += 1
    return n


def dist(a, b, n):
    w = 0.0
    for x, y in zip(a[:n], b[:n]):
        for k in ("sigma_sq", "err_old", "err_new"):
            if not (np.isnan(x[k]) and np.isnan(y[k])):
                w = max(w, abs(x[k] - y[k]) / max(abs(y[k]), 1e-300))
    return w


rows = []
for i in range(int(args[1]) if len(args) > 1 else 150):
    n_cams = int(rng.integers(60, 220)) if big else int(rng.integers(2, 90)); n_pts = int(rng.integers(200, 700)) if big else int(rng.integers(3, 900))
    window = (int(rng.integers(4, 60)) if big else (None if rng.random() < 0.35 else int(rng.integers(2, max(3, n_cams)))))
    case = dict(n_cams=n_cams, n_pts=n_pts, seed=5000 + i, window=window, n_fixed=int(rng.integers(1, min(4, n_cams))),
                outlier_frac=float(rng.choice([0.0, 0.02, 0.15])), pt_noise=float(rng.choice([0.002, 0.01, 0.05])),
                dup=int(rng.choice([1, 1, 1, 3])))
    prob = synth.make_ba_problem(**case)
    if len(prob["cam_idx"]) == 0: continue
    est = [_abi.EST_TUKEY, _abi.EST_CAUCHY, _abi.EST_HUBER][i % 3]
    mi = int(rng.choice([20, 20, 3, 7]))
    a = util.run_ba(hip, prob, estimator=est, max_iterations=mi)
    b = util.run_ba(oracle, prob, estimator=est, max_iterations=mi)
    try:
        util.assert_ba_equal(a, b, rel=1e-6)
        continue
    except AssertionError:
        pass
    r = referee_lib.run_ba(prob, estimator=est, max_iterations=mi)
    n3 = min(common_prefix(a["trials"], r["trials"]), common_prefix(b["trials"], r["trials"]))
    dp, do = dist(a["trials"], r["trials"], n3), dist(b["trials"], r["trials"], n3)
    full = n3 == len(r["trials"]) == len(a["trials"]) == len(b["trials"])
    row = {"i": i, "case": case, "estimator": est, "max_iterations": mi, "trials": [len(a["trials"]), len(b["trials"]), len(r["trials"])],
           "common_prefix": n3, "product_vs_referee": dp, "oracle_vs_referee": do}
    if full:
        row["state_product_vs_referee"] = float(max(np.nanmax(np.abs(a["poses"] - r["poses"])), np.nanmax(np.abs(a["points"] - r["points"]))))
        row["state_oracle_vs_referee"] = float(max(np.nanmax(np.abs(b["poses"] - r["poses"])), np.nanmax(np.abs(b["points"] - r["points"]))))
    # where the product leaves the referee's trajectory and where the oracle does
    row["product_follows_referee_for"] = common_prefix(a["trials"], r["trials"])
    row["oracle_follows_referee_for"] = common_prefix(b["trials"], r["trials"])
    rows.append(row)
    print("CASE", json.dumps(row))
print("done: %d off-tolerance cases" % len(rows))
