#!/bin/bash
# A/B of two builds of the library on the bench's per-kernel times and the accepted trial, alternating in one GPU call.
# usage (GPU box): bash tools/dev/r05_lib_ab.sh <variant under tools/_exp> [reps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in $(seq ${2:-3}); do
for lib in $R/ptam_cg_amd/csrc/libptam_hip.so $R/tools/_exp/$1/libptam_hip.so; do
  PTAM_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-tracking > /tmp/ab_log.txt 2>&1
  python3 - "$lib" <<PY
import json, sys
b = json.loads([l for l in open("/tmp/ab_log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
k = {a: round(1e3 * v, 1) for a, v in b.get("kernel_ms_per_trial", {}).items()}
kl = {a: round(1e3 * v, 1) for a, v in l.get("kernel_ms_per_trial", {}).items()}
print("%-8s accepted trial %.1f us mix %s | %s | local %.1f us sel %.1f prj %.1f | config5 %.0f it/s solve %.1f" % (sys.argv[1].split("/")[-2], b.get("accepted_trial_us", 0), list(b.get("trial_mix", {}).values()), k, 1e3 * l.get("ms_per_step", 0), kl.get("select", 0), kl.get("project", 0), g.get("value", 0), 1e3 * g.get("kernel_ms_per_trial", {}).get("solve", 0)))
PY
done
done
