"""-m gpu: the sharded HIP path on ONE GPU (gpurun gives one device per call):
  - two / three processes share the GPU and exchange through gloo (device buffers staged via the
    ABI's upload/download) — the same kernels, hooks and protocol the 8-GPU run uses, minus RCCL;
  - the built-in RCCL hook is exercised as a 1-rank communicator."""
import ctypes as C

import numpy as np
import pytest

from ptam_cg_amd import _abi, host
from tests import dist_util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,case", [(2, dict(n_cams=10, n_pts=160, seed=5)),
                                        (3, dict(n_cams=24, n_pts=900, seed=6, window=8, n_fixed=2))])
def test_sharded_hip_matches_single_process_oracle(oracle, world, case):
    res = dist_util.run_sharded("hip", world, case)
    dist_util.check_sharded_equals_single(res, oracle, case, rel=1e-6)


def test_sharded_config5_shape_three_ranks(oracle):
    """BASELINE.json configs[4] at more than half its size — 200 keyframes x 26 000 points, 16-camera covisibility window
    (M ~ 0.42 M), camera system 1194 x 1194 with a 4-block band — sharded over three processes on the one GPU: the looping
    K7 form, the banded LDL^T and the band-only exchange of S|E, trial by trial against the single-process oracle"""
    case = dict(n_cams=200, n_pts=26000, seed=0x5EED0006, window=16)
    res = dist_util.run_sharded("hip", 3, case, opts=dict(max_iterations=3), timeout=900)
    dist_util.check_sharded_equals_single(res, oracle, case, rel=1e-6, max_iterations=3)
    assert len(res["trials"]) == 3


def test_abort_raised_on_one_rank_stops_every_rank_at_the_same_trial():
    """the abort byte is rank-local, the trials are collective.  Rank 1 alone raises it (before Compute(), and
    after its 12th all-reduce): every rank must leave the LM loops at the same trial, none may hang"""
    case = dict(n_cams=10, n_pts=160, seed=5)
    res = dist_util.run_sharded("hip", 2, case, abort=(1, 0), timeout=120)
    assert res["trials_accepted_all"] == [(0, 0), (0, 0)]
    res = dist_util.run_sharded("hip", 2, case, abort=(1, 12), timeout=120)
    (n0, a0), (n1, a1) = res["trials_accepted_all"]
    assert n0 == n1 and a0 == a1 and 0 < n0 < 20
    for p in res["poses_all"]:
        assert (p == res["poses"]).all()


def test_prepare_error_on_one_rank_is_returned_by_every_rank():
    """rank 0's shard holds a duplicated (camera, point) measurement — the one input a shard's prepare refuses —,
    rank 1's shard is fine.  Rank 1 must not enter the first all-reduce alone: both return an error, rank 0 its own, rank 1
    the collective one.  (Until round 3 this test used a point seen by 300 cameras, which is no longer refused.)"""
    res = dist_util.run_sharded("hip", 2, dict(n_cams=10, n_pts=160, seed=5), dup=4, timeout=180)
    assert "error" in res and len(res["error"]) == 2
    assert "duplicate" in res["error"][0] and "could not prepare" in res["error"][1]


def test_sharded_long_points_and_many_fixed_cameras(oracle):
    """every point seen by 300 cameras, 290 of them fixed, sharded over two processes: the long-point chunks and the
    free-cameras-first row order on the sharded path"""
    case = dict(n_cams=300, n_pts=24, seed=34, n_fixed=290)
    res = dist_util.run_sharded("hip", 2, case, opts=dict(max_iterations=3))
    dist_util.check_sharded_equals_single(res, oracle, case, rel=1e-6, max_iterations=3)


def test_sharded_select_with_ties_and_slot_overflow(oracle):
    """every point replicated 24 times: bit-identical errors in runs of 24.  With 6-key exchange slots the
    last stage of the sharded select overflows, the host repeats the step on the gather-everything path,
    and the result must still equal the single-process run; with the default slots the ties fit."""
    case = dict(n_cams=8, n_pts=40, seed=12, dup=24)
    for env in ({"PTAM_XCAND_CAP": "6"}, None):
        res = dist_util.run_sharded("hip", 2, case, extra_env=env)
        dist_util.check_sharded_equals_single(res, oracle, case, rel=1e-6)


def test_sharded_rank_without_measurements_is_refused_by_all_ranks():
    """two points over three ranks: rank 2 owns nothing.  It must not drop out of the collectives silently (the others
    would wait for ever): every rank returns the same error"""
    res = dist_util.run_sharded("hip", 3, dict(n_cams=4, n_pts=2, seed=1), timeout=120)
    assert "error" in res and len(res["error"]) == 3 and all("no measurement" in e for e in res["error"])


def test_rccl_single_rank_allreduce(hip):
    ctx = host.Context(lib=hip)
    ident = (C.c_uint8 * 128)()
    ctx._check(hip.rccl_unique_id(ident), "rccl_unique_id")
    comm = C.c_void_p()
    ctx._check(hip.rccl_create(ctx.h, ident, 0, 1, C.byref(comm)), "rccl_create")
    x = np.random.default_rng(0).normal(size=4097)
    d = C.c_void_p()
    ctx._check(hip.dev_alloc(ctx.h, x.nbytes, C.byref(d)), "alloc")
    ctx._check(hip.dev_upload(ctx.h, d, x.ctypes.data, x.nbytes), "upload")
    ctx._check(hip.rccl_allreduce_f64(comm, C.cast(d, C.POINTER(C.c_double)), len(x), hip.ctx_stream(ctx.h)), "allreduce")
    y = np.zeros_like(x)
    ctx._check(hip.dev_download(ctx.h, y.ctypes.data, d, x.nbytes), "download")
    assert np.array_equal(x, y)                     # sum over one rank = identity
    hip.dev_free(ctx.h, d)
    hip.rccl_destroy(comm)


def test_bench_n2_path_prints_one_json_line():
    """bench.py's N > 1 code (rank bookkeeping, MAX over ranks, the one-device run of the same problem in the
    same line) must have executed before the driver's first multi-GPU run does it.  Two ranks under torch.distributed.run,
    both on the one GPU (PTAM_BENCH_ONE_GPU=1: gloo with host staging instead of RCCL — the numbers mean nothing)."""
    import json
    import os
    import subprocess
    import sys
    env = dict(os.environ, PTAM_BENCH_ONE_GPU="1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(dist_util.free_port()), os.path.join(dist_util.ROOT, "bench.py"), "--gpus", "2", "--cams", "24",
           "--points", "900", "--window", "8", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-tracking"]
    r = subprocess.run(cmd, env=env, cwd=dist_util.ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # stdout = exactly the record
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["workload"] == "bundle_24kf_x_900pts_window8_strong_x2"
    one = rec["single_gpu_same_workload"]
    assert one["workload"] == "bundle_24kf_x_900pts_window8" and one["value"] > 0
    assert abs(rec["speedup_vs_single_gpu"] - rec["value"] / one["value"]) < 1e-12
    assert set(rec["kernel_ms_per_trial"]) >= {"jacobian", "schur", "solve"}


def test_solve_fault_on_one_rank_is_repeated_by_every_rank(oracle):
    """a rank whose persistent camera solve gives up a wait (spin limit of one look on rank 1 only) must not
    leave the others in the next all-reduce.  The fault rides in the trial's scalar exchange; every rank repeats the trial with
    the launch-per-block-column form and the adjustment ends as the single-process oracle's does."""
    case = dict(n_cams=64, n_pts=900, seed=43)
    res = dist_util.run_sharded("hip", 3, case, opts=dict(max_iterations=6), rank_env={1: {"PTAM_CH_SPIN_LIMIT": "1"}})
    dist_util.check_sharded_equals_single(res, oracle, case, rel=1e-6, max_iterations=6)
    assert res["solve_fallbacks_all"] == [1, 1, 1]
