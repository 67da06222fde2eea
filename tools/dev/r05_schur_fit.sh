#!/bin/bash
# round 5 (second session): the Schur split's calibrated cost model — bundle parity tests with the product build, A/B of the
# model's knobs with the measurement build (make ab), and the per-workgroup stamps of the new default for a re-fit.
# usage (GPU box): bash tools/dev/r05_schur_fit.sh <tag> "ENV=.. ENV=.." "..." (each argument one configuration)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=${1:-fit1}; shift
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bundle" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for rep in 1 2; do
for cfg in "$@"; do
  env PTAM_HIP_LIB=$R/tools/_ab/libptam_hip.so $cfg timeout 300 python bench.py --no-cpu-baseline --no-tracking > $O/log.txt 2>&1
  python3 - "$cfg" <<PY | tee -a $O/out.txt
import json, sys
b = json.loads([l for l in open("$O/log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
s = lambda d: 1e3 * d.get("kernel_ms_per_trial", {}).get("schur", 0)
print("%-70s schur us: headline %.1f  config5 %.1f  config4 %.1f | it/s %.0f %.0f %.0f | accepted trial %.1f us, mix %s" % (sys.argv[1], s(b), s(g), s(l), b["value"], g.get("value", 0), l.get("value", 0), b.get("accepted_trial_us", 0), list(b.get("trial_mix", {}).values())))
PY
done
done
PTAM_HIP_LIB=$R/tools/_exp/schur_stamps/libptam_hip.so timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-global --no-local --steps 6 --warmup 1 --jac-reps 5 > $O/stamps_log.txt 2>&1
python3 tools/dev/schur_fit.py $O/stamps_log.txt | tee $O/fit.txt
