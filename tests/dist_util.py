import os
import pickle
import socket
import subprocess
import sys
import tempfile

import numpy as np

from ptam_cg_amd import synth
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_sharded(which, world, case, timeout=300, extra_env=None, opts=None, drop=None, abort=None, dup=None, rank_env=None):
    """launch `world` worker processes (gloo on 127.0.0.1); returns rank 0's result dict"""
    out = tempfile.mktemp(suffix=".pkl")
    port = str(free_port())
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   PTAM_DIST_CASE=repr(case), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if opts is not None:
            env["PTAM_DIST_OPTS"] = repr(opts)
        if drop is not None:
            env["PTAM_DIST_DROP"] = repr(drop)
        if abort is not None:
            env["PTAM_DIST_ABORT"] = repr(abort)
        if dup is not None:
            env["PTAM_DIST_DUP"] = repr(dup)
        env.update(extra_env or {})
        env.update((rank_env or {}).get(r, {}))   # (environment of ONE rank: a fault injected there only)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), which, out],
                                      env=env, cwd=ROOT))
    try:
        for p in procs:
            assert p.wait(timeout=timeout) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    with open(out, "rb") as f:
        res = pickle.load(f)
    os.unlink(out)
    return res


def check_sharded_equals_single(sharded, single_lib, case, rel=1e-8, **opts):
    """the N-shard run must reproduce the one-process run trial by trial (sums are re-associated
    across shards, hence a tolerance instead of bit equality)"""
    prob = synth.make_ba_problem(**case)
    one = util.run_ba(single_lib, prob, **opts)
    ts, to = sharded["trials"], one["trials"]
    assert len(ts) == len(to), (len(ts), len(to))
    assert np.array_equal(ts["lambda"], to["lambda"]) and np.array_equal(ts["accepted"], to["accepted"])
    assert np.array_equal(ts["n_bad"], to["n_bad"])
    for k in ("sigma_sq", "err_old", "err_new", "sum_sq_update"):
        assert np.allclose(ts[k], to[k], rtol=rel, atol=1e-15), k
    assert sharded["accepted"] == one["accepted"] and sharded["converged"] == one["converged"]
    assert np.allclose(sharded["poses"], one["poses"], atol=1e-9)
    for p in sharded["poses_all"]:
        assert np.allclose(p, sharded["poses"], atol=1e-12)          # every rank holds the same cameras
    assert np.allclose(sharded["points"], one["points"], atol=1e-9)
    so = {tuple(x) for x in np.asarray(sharded["outliers"]).reshape(-1, 2)}
    oo = {tuple(x) for x in np.asarray(one["outliers"]).reshape(-1, 2)}
    assert so == oo
