#!/bin/bash
# round 6: the even split of the Schur work lists (default) against the greedy fill + budget search (PTAM_SPLIT_GREEDY=1, measurement
# build), alternating in one call: Schur us per trial at the headline, config 5 and the local bundle, and prepare_ms.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
export PTAM_HIP_LIB=$R/tools/_ab/libptam_hip.so
OUT=$R/gpurun_out/r06_split_ab.txt
: > $OUT
for rep in 1 2 3; do
for gr in 0 1; do
  if [ $gr = 1 ]; then export PTAM_SPLIT_GREEDY=1; else unset PTAM_SPLIT_GREEDY; fi
  timeout 300 python bench.py --no-cpu-baseline --no-tracking > /tmp/ab_log.txt 2>&1
  python3 - "$gr" >> $OUT <<PY
import json, sys
b = json.loads([l for l in open("/tmp/ab_log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
s = lambda d: 1e3 * d.get("kernel_ms_per_trial", {}).get("schur", 0)
print("greedy=%s schur us: headline %.1f  config5 %.1f  config4 %.1f | it/s %.0f %.0f %.0f | prepare %.3f ms" % (sys.argv[1], s(b), s(g), s(l), b["value"], g.get("value", 0), l.get("value", 0), b["prepare_ms"]))
PY
done
done
cat $OUT
