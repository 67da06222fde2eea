"""the looping form of K7 (more chunks than resident wave slots) on points measured by more than 64 cameras: HIP vs oracle"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util
hip, oracle = load(), load_oracle()
p = synth.make_ba_problem(100, 5000, 51)
print("M", len(p["cam_idx"]), flush=True)
a = util.run_ba(hip, p, max_iterations=3); b = util.run_ba(oracle, p, max_iterations=3)
util.assert_ba_equal(a, b, rel=1e-6); print("100 x 5000 dense: EQUAL", len(a["trials"]), "trials")
