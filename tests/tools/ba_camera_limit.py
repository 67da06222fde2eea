import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import synth, host
from ptam_cg_amd._lib import load
from tests import util
from tests.oracle_lib import load_oracle
hip = load()
# 400 cameras against the oracle (narrower K7 workgroups: the LDS partials of 399 free cameras leave room for 8 waves)
p = synth.make_ba_problem(400, 1200, 43, window=6)
a, b = util.run_ba(hip, p, max_iterations=3), util.run_ba(load_oracle(), p, max_iterations=3)
util.assert_ba_equal(a, b, rel=1e-6); print("400 cameras vs oracle: EQUAL", len(a["trials"]), "trials")
for cams in (400, 500, 700, 1000):
    p = synth.make_ba_problem(cams, 2000, 41, window=6)
    try:
        a = util.run_ba(hip, p, max_iterations=3)
        print(cams, "cameras: ok, trials", len(a["trials"]), "accepted", a["accepted"], "err", float(a["trials"]["err_new"][-1]))
    except Exception as e:
        print(cams, "cameras:", repr(e)[:200])
