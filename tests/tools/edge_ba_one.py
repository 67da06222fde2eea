import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from ptam_cg_amd import _abi, host, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util
which = sys.argv[1]
import faulthandler; faulthandler.dump_traceback_later(12, exit=True)
hip, oracle = load(), load_oracle()
p = synth.make_ba_problem(6, 40, 3)
sub = lambda keep: {k: (v[keep] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in p.items()}
kw = dict(max_iterations=6)
if which == "allfixed": q = dict(p); q["fixed"] = np.ones_like(p["fixed"])
elif which == "nofixed": q = dict(p); q["fixed"] = np.zeros_like(p["fixed"])
elif which == "onept":
    q = sub(p["pt_idx"] == 0); q["points"] = p["points"][:1]; q["points_true"] = p["points_true"][:1]
elif which == "onemeas":
    q = sub(p["pt_idx"] == 0); q["points"] = p["points"][:1]; q["points_true"] = p["points_true"][:1]
    q = {k: (v[:1] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in q.items()}
elif which == "unobserved": q = sub((p["pt_idx"] != 5) & (p["cam_idx"] != 3))
elif which == "fixedonly": q = sub(~((p["pt_idx"] < 5) & (p["cam_idx"] != 0)))
elif which == "c65": q = synth.make_ba_problem(65, 30, 8)
elif which == "c64": q = synth.make_ba_problem(64, 30, 8)
elif which == "outl": q = synth.make_ba_problem(8, 200, 9, outlier_frac=0.45)
b = util.run_ba(oracle, q, **kw)
print(which, "oracle: trials", len(b["trials"]), "acc", b["accepted"], "outl", len(b["outliers"]), flush=True)
a = util.run_ba(hip, q, **kw)
print(which, "hip   : trials", len(a["trials"]), "acc", a["accepted"], "outl", len(a["outliers"]), flush=True)
try:
    util.assert_ba_equal(a, b, rel=1e-6); print(which, "EQUAL", flush=True)
except AssertionError as e:
    print(which, "DIFF", str(e)[:200], flush=True)
