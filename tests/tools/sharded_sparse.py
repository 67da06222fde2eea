import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util, dist_util
hip, oracle = load(), load_oracle()
case = dict(n_cams=30, n_pts=3000, seed=7, window=2)
p = synth.make_ba_problem(**case)
a, b = util.run_ba(hip, p), util.run_ba(oracle, p)
ta, tb = a["trials"], b["trials"]
print("unsharded: trials", len(ta), len(tb), "lambda eq", np.array_equal(ta["lambda"], tb["lambda"]), "acc eq", np.array_equal(ta["accepted"], tb["accepted"]))
print(" rel diff err_new per trial:", np.abs(ta["err_new"] - tb["err_new"]) / np.abs(tb["err_new"]))
res = dist_util.run_sharded("hip", 3, case)
ts = res["trials"]
print("sharded  : trials", len(ts), "lambda eq", np.array_equal(ts["lambda"], tb["lambda"]))
print(" rel diff err_new per trial:", np.abs(ts["err_new"] - tb["err_new"]) / np.abs(tb["err_new"]))
print(" rel diff err_old per trial:", np.abs(ts["err_old"] - tb["err_old"]) / np.abs(tb["err_old"]))
