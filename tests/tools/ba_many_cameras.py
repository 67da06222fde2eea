"""camera counts beyond the committed cases (long banded chains, wide dense systems): HIP vs oracle"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util
hip, oracle = load(), load_oracle()
for (cams, pts, kw) in ((300, 3000, dict(window=8)), (120, 400, {}), (257, 1500, dict(window=40)), (33, 90, {})):
    p = synth.make_ba_problem(cams, pts, 41, **kw)
    a, b = util.run_ba(hip, p, max_iterations=5), util.run_ba(oracle, p, max_iterations=5)
    try:
        util.assert_ba_equal(a, b, rel=1e-6); print(cams, pts, kw, "M", len(p["cam_idx"]), "EQUAL", len(a["trials"]), "trials, accepted", a["accepted"])
    except AssertionError as e:
        print(cams, pts, kw, "DIFF", str(e)[:160])
