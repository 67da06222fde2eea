#!/bin/bash
# round 6: the trial's decision inside the step-closing launch (default) against a launch of its own (PTAM_SEPARATE_FINALIZE=1,
# measurement build), alternating in one call: per-trial times by outcome at the headline, the local bundle's per-trial time.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
export PTAM_HIP_LIB=$R/tools/_ab/libptam_hip.so
OUT=$R/gpurun_out/r06_decide_ab.txt
: > $OUT
for rep in 1 2 3; do
for sep in 0 1; do
  if [ $sep = 1 ]; then export PTAM_SEPARATE_FINALIZE=1; else unset PTAM_SEPARATE_FINALIZE; fi
  echo "separate_finalize=$sep: $(python tools/dev/r06_trial_times.py 12 2>&1 | head -1)" >> $OUT
  timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-global > /tmp/ab_log.txt 2>&1
  python3 - "$sep" >> $OUT <<PY
import json, sys
b = json.loads([l for l in open("/tmp/ab_log.txt") if l.startswith("{")][-1])
l = b.get("local_ba_config4", {})
print("separate_finalize=%s: value %.0f mix %s | local %.0f it/s, five calls %s, accepted %.1f us" % (sys.argv[1], b["value"], list(b["trial_mix"].values()), l.get("value", 0), l.get("ms_per_step_of_5_calls"), l.get("accepted_trial_us", 0)))
PY
done
done
cat $OUT
