#!/bin/bash
# round 5: A/B of the two-phase Schur complement (measurement build: make -C ptam_cg_amd/csrc ab).  Per-kernel events switch the
# two-phase form off, so the figure is the accepted trial and the bench value.  usage (GPU box): bash tools/dev/r05_two.sh <tag> "ENV=.." ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-two1}; shift
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bundle" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for rep in 1 2; do
for cfg in "$@"; do
  env PTAM_HIP_LIB=$R/tools/_ab/libptam_hip.so $cfg timeout 300 python bench.py --no-cpu-baseline --no-tracking > $O/log.txt 2> $O/err.txt
  grep "two-phase" $O/err.txt | head -3
  python3 - "$cfg" <<PY | tee -a $O/out.txt
import json, sys
b = json.loads([l for l in open("$O/log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
print("%-45s it/s %.0f %.0f %.0f | accepted trial %.1f us, mix %s | local %.1f us | cold %.0f | det %.0f" % (sys.argv[1], b["value"], g.get("value", 0), l.get("value", 0), b.get("accepted_trial_us", 0), list(b.get("trial_mix", {}).values()), 1e3 * l.get("ms_per_step", 0), b.get("cold_call", {}).get("value", 0), b.get("deterministic_mode", {}).get("value", 0)))
PY
done
done
