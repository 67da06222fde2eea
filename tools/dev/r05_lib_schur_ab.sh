#!/bin/bash
# A/B of two builds of the library on the bench's Schur time, alternating in one GPU call.  usage (GPU box): bash tools/dev/r05_lib_schur_ab.sh <variant under tools/_exp> [reps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in $(seq ${2:-3}); do
for lib in $R/ptam_cg_amd/csrc/libptam_hip.so $R/tools/_exp/$1/libptam_hip.so; do
  PTAM_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-tracking > /tmp/ab_log.txt 2>&1
  python3 - "$lib" <<PY
import json, sys
b = json.loads([l for l in open("/tmp/ab_log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
s = lambda d: 1e3 * d.get("kernel_ms_per_trial", {}).get("schur", 0)
print("%-14s schur us: headline %.1f  config5 %.1f  config4 %.1f | accepted trial %.1f us" % (sys.argv[1].split("/")[-2], s(b), s(g), s(l), b.get("accepted_trial_us", 0)))
PY
done
done
