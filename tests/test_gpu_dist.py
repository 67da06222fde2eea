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
