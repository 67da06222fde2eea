#!/bin/bash
# round 6: select_compact_kernel touching K7's inputs (default) against PTAM_NO_K7_PREFETCH=1 (measurement build), alternating in one
# call: per-trial times by outcome, K7 / select per trial from the profiled run.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
export PTAM_HIP_LIB=$R/tools/_ab/libptam_hip.so
OUT=$R/gpurun_out/r06_prefetch_ab.txt
: > $OUT
for rep in 1 2 3; do
for off in 0 1; do
  if [ $off = 1 ]; then export PTAM_NO_K7_PREFETCH=1; else unset PTAM_NO_K7_PREFETCH; fi
  echo "no_prefetch=$off: $(python tools/dev/r06_trial_times.py 12 2>&1 | head -1)" >> $OUT
  timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-global --no-local > /tmp/ab_log.txt 2>&1
  python3 - "$off" >> $OUT <<PY
import json, sys
b = json.loads([l for l in open("/tmp/ab_log.txt") if l.startswith("{")][-1])
k = b["kernel_ms_per_trial"]
print("no_prefetch=%s: value %.0f mix %s | profiled: jacobian %.1f select %.1f us | K7 in compute frac %.3f" % (sys.argv[1], b["value"], list(b["trial_mix"].values()), 1e3 * k["jacobian"], 1e3 * k["select"], b["roofline"].get("frac_in_compute", 0)))
PY
done
done
cat $OUT
