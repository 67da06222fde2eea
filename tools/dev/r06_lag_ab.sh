#!/bin/bash
# round 6: what the CU's second workgroup is given less than its first (PTAM_SCHUR_LAG, measurement build) — Schur us per trial at
# the headline, config 5 and the local bundle.   -> gpurun_out/r06_lag_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
OUT=$R/gpurun_out/r06_lag_ab.txt
: > $OUT
for rep in 1 2; do
for lag in "$@"; do
  env PTAM_HIP_LIB=$R/tools/_ab/libptam_hip.so PTAM_SCHUR_LAG=$lag timeout 300 python bench.py --no-cpu-baseline --no-tracking > /tmp/ab_log.txt 2>&1
  python3 - "$lag" >> $OUT <<PY
import json, sys
b = json.loads([l for l in open("/tmp/ab_log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
s = lambda d: 1e3 * d.get("kernel_ms_per_trial", {}).get("schur", 0)
print("lag %-6s schur us: headline %.1f  config5 %.1f  config4 %.1f | it/s %.0f %.0f %.0f" % (sys.argv[1], s(b), s(g), s(l), b["value"], g.get("value", 0), l.get("value", 0)))
PY
done
done
cat $OUT
