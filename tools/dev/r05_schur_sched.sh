#!/bin/bash
# round 5: A/B of the Schur work split (measurement build: make -C ptam_cg_amd/csrc ab): per-leg Schur time of the bench (HIP events,
# tile + reduce) for a list of env settings.  usage (GPU box): bash tools/dev/r05_schur_sched.sh "A=1" "PTAM_SCHUR_SEGCOST=200" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export PTAM_HIP_LIB=${PTAM_HIP_LIB:-$R/tools/_ab/libptam_hip.so}
O=$R/gpurun_out/r05_schur_sched; mkdir -p $O
cd $R
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --no-cpu-baseline --no-tracking > $O/log.txt 2>&1
  python3 - "$cfg" <<PY | tee -a $O/out.txt
import json, sys
b = json.loads([l for l in open("$O/log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
s = lambda d: 1e3 * d.get("kernel_ms_per_trial", {}).get("schur", 0)
print("%-40s schur us: headline %.1f  config5 %.1f  config4 %.1f | it/s %.0f %.0f %.0f | accepted trial %.1f us, mix %s" % (sys.argv[1], s(b), s(g), s(l), b["value"], g.get("value", 0), l.get("value", 0), b.get("accepted_trial_us", 0), list(b.get("trial_mix", {}).values())))
PY
done
done
