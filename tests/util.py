"""Shared helpers for parity tests: run the same scenario through two bound libraries."""
import numpy as np

from ptam_cg_amd import host, synth


def keyframe_levels(lib, im, variant=0):
    ctx = host.Context(lib=lib, size=(im.shape[1], im.shape[0]), halfsample=variant)
    kf = host.KeyFrame(ctx).MakeKeyFrame_Lite(im)
    out = [kf.level(l) for l in range(4)]
    kf.close()
    ctx.close()
    return out


def assert_levels_equal(a, b):
    for l in range(4):
        assert a[l]["im"].shape == b[l]["im"].shape
        assert np.array_equal(a[l]["im"], b[l]["im"]), f"level {l} pixels differ"
        assert np.array_equal(a[l]["corners"], b[l]["corners"]), f"level {l} corners differ"
        assert np.array_equal(a[l]["rowlut"], b[l]["rowlut"]), f"level {l} row LUT differs"


def run_ba(lib, prob, **opts):
    ctx = host.Context(lib=lib)
    ba = synth.load_into(host.Bundle(ctx, **opts), prob)
    acc = ba.Compute()
    poses, pts = ba.get_all()
    res = {"accepted": acc, "converged": ba.Converged(), "trials": ba.trials(), "poses": poses,
           "points": pts, "outliers": ba.GetOutlierMeasurements(), "solve_fallbacks": ba.solve_fallbacks()}
    ba.close()
    ctx.close()
    return res


def assert_ba_equal(a, b, rel=1e-6, abs_state=1e-7):
    """trial-by-trial comparison (compare lambda, sigma^2, cur, new, n_bad per trial)"""
    ta, tb = a["trials"], b["trials"]
    assert len(ta) == len(tb), (len(ta), len(tb))
    for i, (x, y) in enumerate(zip(ta, tb)):
        assert x["lambda"] == y["lambda"], (i, x["lambda"], y["lambda"])
        assert x["accepted"] == y["accepted"], i
        assert x["n_bad"] == y["n_bad"], (i, x["n_bad"], y["n_bad"])
        for k in ("sigma_sq", "err_old", "err_new"):
            if np.isnan(x[k]) and np.isnan(y[k]):
                continue   # a rank-deficient camera system (more free cameras than the points constrain): NaN on both sides
            assert abs(x[k] - y[k]) <= rel * max(abs(x[k]), abs(y[k]), 1e-300), (i, k, x[k], y[k])
    assert a["accepted"] == b["accepted"]
    assert a["converged"] == b["converged"]
    assert np.array_equal(a["outliers"], b["outliers"])
    assert np.allclose(a["poses"], b["poses"], rtol=0, atol=abs_state, equal_nan=True)
    assert np.allclose(a["points"], b["points"], rtol=0, atol=abs_state, equal_nan=True)


def run_ba_subprocess(case, env=None, opts=None, timeout=600):
    """the product library on synth.make_ba_problem(**case) in a process of its own (environment switches of the library are read
    once per process) -> the result dict of run_ba"""
    import os
    import pickle
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tempfile.mktemp(suffix=".pkl")
    code = ("import sys, pickle; sys.path.insert(0, %r)\n"
            "from ptam_cg_amd import synth\nfrom ptam_cg_amd._lib import load\nfrom tests import util\n"
            "res = util.run_ba(load(), synth.make_ba_problem(**%r), **%r)\n"
            "pickle.dump(res, open(%r, 'wb'))\n") % (root, case, opts or {}, out)
    e = dict(os.environ)
    e.update(env or {})
    subprocess.run([sys.executable, "-c", code], env=e, cwd=root, check=True, timeout=timeout)
    with open(out, "rb") as f:
        res = pickle.load(f)
    os.unlink(out)
    return res
