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
