"""Worker of the world_size-2 sharded bundle test.  Backend library: 'oracle' (CPU, gloo — runs in
the build container) or 'hip' (two processes sharing one GPU, gloo with host staging — GPU box)."""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    which, out_path = sys.argv[1], sys.argv[2]
    import numpy as np
    import torch  # noqa: F401
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}",
                            rank=rank, world_size=world)
    from ptam_cg_amd import host, synth
    from ptam_cg_amd.sharding import gather_points, merge_outliers, shard_problem, torch_allreduce_hook
    if which == "oracle":
        from tests.oracle_lib import load_oracle
        lib = load_oracle()
    else:
        from ptam_cg_amd._lib import load
        lib = load()
    kw = eval(os.environ.get("PTAM_DIST_CASE", "dict(n_cams=10, n_pts=160, seed=5)"))
    opts = eval(os.environ.get("PTAM_DIST_OPTS", "dict()"))                   # host.Bundle options (max_iterations ...)
    drop = eval(os.environ.get("PTAM_DIST_DROP", "None"))                      # (point modulus, residue, first camera dropped)
    dup = eval(os.environ.get("PTAM_DIST_DUP", "None"))                        # global point id whose first measurement is added twice
    abort_at = eval(os.environ.get("PTAM_DIST_ABORT", "None"))                 # (rank, all-reduce calls before the flag goes up)
    prob = synth.make_ba_problem(**kw)
    if drop is not None:     # thin out some points' measurements: lets ONE shard break a per-point limit
        mod, res, cam0 = drop
        keep = ~((prob["pt_idx"] % mod == res) & (prob["cam_idx"] >= cam0))
        for k in ("cam_idx", "pt_idx", "found", "sigma_sq"):
            prob[k] = prob[k][keep]
    if dup is not None:      # a duplicated (camera, point) measurement: the one input a shard's prepare refuses (PTAM_E_ARG)
        i = int(np.nonzero(prob["pt_idx"] == dup)[0][0])
        for k in ("cam_idx", "pt_idx", "found", "sigma_sq"):
            prob[k] = np.concatenate([prob[k], prob[k][i:i + 1]])
    mine = shard_problem(prob, rank, world)
    ctx = host.Context(lib=lib)
    ba = synth.load_into(host.Bundle(ctx, **opts), mine)
    inner = torch_allreduce_hook(ctx, device_ptr=(which == "hip"))
    abort = np.zeros(1, dtype=np.uint8)
    calls = [0]
    if abort_at is not None and abort_at[0] == rank:
        from ptam_cg_amd import _abi
        if abort_at[1] == 0:
            abort[0] = 1

        def counting(user, ptr, count, stream):   # the flag of THIS rank only goes up in the middle of the run
            calls[0] += 1
            if calls[0] >= abort_at[1]:
                abort[0] = 1
            return inner(user, ptr, count, stream)
        hook = _abi.ALLREDUCE_FN(counting)
    else:
        hook = inner
    ba.set_comm(rank, world, hook)
    try:
        acc = ba.Compute(abort)
    except host.PtamError as e:          # (every rank raises together: the refusal is decided by a collective)
        errs = [None] * world
        dist.all_gather_object(errs, str(e))
        if rank == 0:
            with open(out_path, "wb") as f:
                pickle.dump(dict(error=errs), f)
        dist.barrier()
        dist.destroy_process_group()
        return
    poses, pts = ba.get_all()
    n_trials_all = [None] * world
    dist.all_gather_object(n_trials_all, (len(ba.trials()), acc))
    fallbacks_all = [None] * world
    dist.all_gather_object(fallbacks_all, ba.solve_fallbacks())

    def all_gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    full_pts = gather_points(pts, mine["global_point_ids"], len(prob["points"]), all_gather)
    outl = merge_outliers(ba.GetOutlierMeasurements(), mine["global_point_ids"], all_gather)
    all_poses = all_gather(poses)
    if rank == 0:
        with open(out_path, "wb") as f:
            pickle.dump(dict(accepted=acc, converged=ba.Converged(), trials=ba.trials(), poses=poses, points=full_pts,
                             outliers=outl, poses_all=all_poses, trials_accepted_all=n_trials_all, solve_fallbacks_all=fallbacks_all), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
