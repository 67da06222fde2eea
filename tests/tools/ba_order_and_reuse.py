"""bundle inputs the committed cases do not cover: measurements added in random order; Compute() called again on the same
bundle after a run that purged outliers (HIP vs oracle, trial by trial)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
from tests import util
hip, oracle = load(), load_oracle()
rng = np.random.default_rng(3)
for (cams, pts, kw) in ((12, 500, {}), (30, 800, dict(window=6)), (9, 300, dict(outlier_frac=0.15))):
    p = synth.make_ba_problem(cams, pts, 31, **kw)
    perm = rng.permutation(len(p["cam_idx"]))
    q = {k: (v[perm] if k in ("cam_idx", "pt_idx", "found", "sigma_sq") else v) for k, v in p.items()}
    a, b = util.run_ba(hip, q), util.run_ba(oracle, q)
    try:
        util.assert_ba_equal(a, b, rel=1e-6); print(cams, pts, kw, "shuffled insertion order: EQUAL, outliers", len(a["outliers"]))
    except AssertionError as e:
        print(cams, pts, kw, "shuffled insertion order: DIFF", str(e)[:160])
    # reuse: two Compute() calls on one bundle
    res = []
    for lib in (hip, oracle):
        ctx = host.Context(lib=lib)
        ba = synth.load_into(host.Bundle(ctx, max_iterations=6), q)
        ba.Compute(); t1 = ba.trials().copy(); o1 = np.array(ba.GetOutlierMeasurements())
        ba.Compute(); t2 = ba.trials().copy(); o2 = np.array(ba.GetOutlierMeasurements())
        poses, ptsv = ba.get_all()
        res.append((t1, o1, t2, o2, poses, ptsv)); ba.close(); ctx.close()
    (t1, o1, t2, o2, po, pv), (u1, v1, u2, v2, qo, qv) = res
    ok = len(t2) == len(u2) and np.array_equal(o2, v2) and np.array_equal(t2["accepted"], u2["accepted"]) and \
        np.allclose(t2["err_new"], u2["err_new"], rtol=1e-6) and np.allclose(po, qo, atol=1e-7) and np.allclose(pv, qv, atol=1e-7)
    print(cams, pts, kw, "second Compute():", "EQUAL" if ok else "DIFF", "trials", len(t2), len(u2), "outliers after", len(o2), len(v2))
