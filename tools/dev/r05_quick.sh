#!/bin/bash
# bundle parity tests + N bench lines (no CPU baseline, no tracking) of the product build in one GPU call.  usage (GPU box): bash tools/dev/r05_quick.sh <tag> [N]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/${1:-quick}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bundle" 2>&1 | tail -2
for rep in $(seq ${2:-3}); do
  timeout 300 python bench.py --no-cpu-baseline --no-tracking > $O/log.txt 2>&1
  python3 - <<PY | tee -a $O/out.txt
import json
b = json.loads([l for l in open("$O/log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
k = {a: round(1e3 * v, 1) for a, v in b.get("kernel_ms_per_trial", {}).items()}
print("it/s %.0f %.0f %.0f | accepted trial %.1f us, mix %s | local %.1f us | %s" % (b["value"], g.get("value", 0), l.get("value", 0), b.get("accepted_trial_us", 0), list(b.get("trial_mix", {}).values()), 1e3 * l.get("ms_per_step", 0), k))
PY
done
