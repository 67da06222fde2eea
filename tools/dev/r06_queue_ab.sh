#!/bin/bash
# round 6: a rejected trial's continuation on the context's second queue (PTAM_TWO_QUEUES=1; the default when this was measured)
# against the one queue, alternating runs of the
# headline leg in one GPU call.   -> gpurun_out/r06_queue_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
OUT=$R/gpurun_out/r06_queue_ab.txt
: > $OUT
for rep in 1 2 3 4; do
for one in 0 1; do
  if [ $one = 1 ]; then unset PTAM_TWO_QUEUES; else export PTAM_TWO_QUEUES=1; fi
  timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-global --no-local > /tmp/ab_log.txt 2>&1
  python3 - "$one" >> $OUT <<PY
import json, sys
b = json.loads([l for l in open("/tmp/ab_log.txt") if l.startswith("{")][-1])
dm = b.get("deterministic_mode", {})
print("one_queue=%s  value %.0f it/s mix %s | deterministic %.0f it/s mix %s | cold %.0f | accepted %.1f us | schur %.1f us" % (
    sys.argv[1], b["value"], list(b["trial_mix"].values()), dm.get("value", 0), list(dm.get("trial_mix", {}).values()),
    b["cold_call"]["value"], b.get("accepted_trial_us", 0), 1e3 * b["kernel_ms_per_trial"]["schur"]))
PY
done
done
cat $OUT
