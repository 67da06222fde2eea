#!/bin/bash
# A/B of the Schur split's knobs on the three bench shapes (measurement build), Schur time per trial from the per-kernel events
# usage (GPU box): bash tools/dev/r05_schur_knobs.sh <tag> <reps> "ENV=.. ENV=.." ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=$1; N=$2; shift 2
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
for rep in $(seq $N); do
for cfg in "$@"; do
  env PTAM_HIP_LIB=$R/tools/_ab/libptam_hip.so $cfg timeout 300 python bench.py --no-cpu-baseline --no-tracking > $O/log.txt 2>&1
  python3 - "$cfg" <<PY | tee -a $O/out.txt
import json, sys
b = json.loads([l for l in open("$O/log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
s = lambda d: 1e3 * d.get("kernel_ms_per_trial", {}).get("schur", 0)
print("%-60s schur us: headline %.1f  config5 %.1f  config4 %.1f | accepted trial %.1f us" % (sys.argv[1], s(b), s(g), s(l), b.get("accepted_trial_us", 0)))
PY
done
done
