"""sub-pixel agreement of the ReFind paths with the oracle / the golden fixtures: the statistic behind golden_util.assert_refind_equal"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from tests import golden_util as G
orig = G.assert_refind_equal
def patched(r, found, level, sub_pix, never_retry, root_pos):
    f = found == 1
    d = np.abs(r["root_pos"][f] - root_pos[f]).max(1) if f.any() else np.zeros(0)
    print("   found %d: max |d| %.3e px, %d positions off by more than 1e-6 px (%.2f %%), more than 1e-9: %d" % (int(f.sum()), d.max() if d.size else 0.0, int((d > 1e-6).sum()),
          100.0 * (d > 1e-6).mean() if d.size else 0.0, int((d > 1e-9).sum())))
    orig(r, found, level, sub_pix, never_retry, root_pos)
G.assert_refind_equal = patched
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
import tests.test_gpu_parity as P
hip, oracle = load(), load_oracle()
print("golden refind"); G.check_refind(hip)
print("golden refind pairs"); G.check_refind_pairs(hip)
print("refind_common vs oracle"); P.test_refind_common_matches_oracle.__wrapped__(hip, oracle) if hasattr(P.test_refind_common_matches_oracle, "__wrapped__") else P.test_refind_common_matches_oracle(hip, oracle)
print("refind pairs vs oracle"); P.test_refind_pairs_through_one_patchfinder_matches_oracle(hip, oracle)
