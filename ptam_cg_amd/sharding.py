"""Host logic of sharded global bundle adjustment.

Measurements are sharded BY POINT: rank r owns the points p with p % world == r and every measurement
of those points; all cameras are replicated.  Everything per point (V, epsB, V*^-1, W, the Schur
products, delta b) is then local; per lambda trial the ranks exchange ONE buffer — the camera system
S|E — by sum-all-reduce, plus two scalar pairs and the gathered squared errors for the exact global
median.  The collective itself is a C callback (`ptam_allreduce_f64_fn`): the library ships an RCCL
implementation (ptam_rccl_*); `torch_allreduce_hook` below adapts torch.distributed (gloo or nccl)."""
import ctypes as C

import numpy as np

from . import _abi


def shard_problem(prob, rank, world):
    """-> the rank's sub-problem (points renumbered 0..n_local-1) + 'global_point_ids'."""
    n_pts = len(prob["points"])
    keep_pt = np.flatnonzero(np.arange(n_pts) % world == rank)
    if world == 1:
        out = dict(prob)
        out["global_point_ids"] = keep_pt
        return out
    remap = -np.ones(n_pts, dtype=np.int64)
    remap[keep_pt] = np.arange(len(keep_pt))
    km = remap[prob["pt_idx"]] >= 0
    out = dict(prob)
    out["points"] = prob["points"][keep_pt]
    if "points_true" in prob:
        out["points_true"] = prob["points_true"][keep_pt]
    out["cam_idx"] = prob["cam_idx"][km]
    out["pt_idx"] = remap[prob["pt_idx"][km]].astype(np.int32)
    out["found"] = prob["found"][km]
    out["sigma_sq"] = prob["sigma_sq"][km]
    out["global_point_ids"] = keep_pt
    out["global_meas_ids"] = np.flatnonzero(km)
    return out


def gather_points(local_points, global_ids, n_total, all_gather):
    """Re-assemble the full point array from the shards.  all_gather(obj) -> list of objs per rank."""
    full = np.zeros((n_total, 3))
    for pts, ids in all_gather((local_points, global_ids)):
        full[ids] = pts
    return full


def merge_outliers(local_outliers, global_point_ids, all_gather):
    """(local point, camera) pairs -> global ids, concatenated in rank order."""
    loc = np.asarray(local_outliers, dtype=np.int64).reshape(-1, 2)
    mine = np.column_stack([global_point_ids[loc[:, 0]], loc[:, 1]]) if len(loc) else np.zeros((0, 2), np.int64)
    return np.concatenate(all_gather(mine))


def torch_allreduce_hook(ctx=None, device_ptr=False, group=None):
    """Adapts torch.distributed.all_reduce to the library's all-reduce callback.

    device_ptr=False: the buffer is host memory (CPU test doubles).
    device_ptr=True : the buffer is device memory of `ctx`; it is staged through the host with
                      ptam_dev_download / ptam_dev_upload (lets two processes that share ONE GPU
                      exercise the sharded HIP path over gloo).  Production multi-GPU runs use the
                      built-in RCCL hook (ptam_rccl_allreduce_f64) instead."""
    import torch
    import torch.distributed as dist

    def _hook(user, ptr, count, stream):
        try:
            if count == 0:
                return 0
            if not device_ptr:
                arr = np.ctypeslib.as_array(ptr, shape=(count,))
                t = torch.from_numpy(arr)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                return 0
            host = np.empty(count, dtype=np.float64)
            addr = C.cast(ptr, C.c_void_p)
            rc = ctx.lib.dev_download(ctx.h, host.ctypes.data, addr, count * 8)
            if rc:
                return rc
            t = torch.from_numpy(host)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return ctx.lib.dev_upload(ctx.h, addr, host.ctypes.data, count * 8)
        except Exception as e:  # never let an exception cross the C boundary
            print("allreduce hook failed:", e)
            return -1

    return _abi.ALLREDUCE_FN(_hook)
