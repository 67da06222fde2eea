"""CPU, world_size 2 and 3 over gloo: the sharded global-BA protocol — shard by point,
all-reduce of the camera system S|E, error scalars and gathered e^2 for the exact global median —
driven through the product's sharding / hook code with the CPU oracle as compute backend."""
import pytest

from tests import dist_util


@pytest.mark.parametrize("world,case", [(2, dict(n_cams=10, n_pts=160, seed=5)),
                                        (3, dict(n_cams=12, n_pts=150, seed=6, window=6, n_fixed=2))])
def test_sharded_oracle_matches_single_process(oracle, world, case):
    res = dist_util.run_sharded("oracle", world, case)
    dist_util.check_sharded_equals_single(res, oracle, case)


def test_abort_raised_on_one_rank_stops_every_rank_at_the_same_trial(oracle):
    """the abort flag (src/Bundle.cc:134,338) is local to a process; sharded, every trial runs collectives, so a flag that
    only ONE rank sees must still end the optimisation on ALL ranks at the same trial (no rank left waiting in an
    all-reduce, no diverging lambda history)"""
    case = dict(n_cams=10, n_pts=160, seed=5)
    full = dist_util.run_sharded("oracle", 2, case)
    n_full = len(full["trials"])
    assert n_full >= 3
    # raised before Compute(): nothing runs anywhere
    res = dist_util.run_sharded("oracle", 2, case, abort=(1, 0), timeout=120)
    assert res["trials_accepted_all"] == [(0, 0), (0, 0)]
    # raised by rank 1 in the middle of the run
    res = dist_util.run_sharded("oracle", 2, case, abort=(1, 12), timeout=120)
    (n0, a0), (n1, a1) = res["trials_accepted_all"]
    assert n0 == n1 and a0 == a1 and 0 < n0 < n_full
    for p in res["poses_all"]:
        assert (p == res["poses"]).all()
