#!/bin/bash
# round 6: the kernels of ptam_ba_prepare (csrc/ba_prepare.inc) one by one — rocprofv3 kernel stats of bundles that are only
# built (no Compute), per shape.   usage: tools/dev/r06_prepare_kernels.sh   -> gpurun_out/r06_prepare_kernels.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/r06_prepare_kernels.txt
for shape in "50 5000 0" "200 50000 16" "20 3000 0"; do
  set -- $shape
  rm -rf /tmp/prep_trace
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prep_trace -o p -- python - "$@" > /tmp/prep_trace.log 2>&1 <<PY
import sys, os, time
sys.path.insert(0, "$R")
import numpy as np
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
cams, pts, win = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) or None
ctx = host.Context(lib=load())
prob = synth.make_ba_problem(cams, pts, 11, window=win)
for rep in range(20):
    ba = synth.load_into(host.Bundle(ctx), prob)
    ba.prepare(); ctx.sync(); ba.close()
PY
  echo "== $1 x $2 window $3: 20 prepares" >> $OUT/r06_prepare_kernels.txt
  python3 - >> $OUT/r06_prepare_kernels.txt <<PY
import csv, glob
f = glob.glob("/tmp/prep_trace/**/p_kernel_stats.csv", recursive=True)
tot = 0.0
for r in csv.DictReader(open(f[0])):
    n = r["Name"].split("(")[0]
    calls = int(r["Calls"])
    if calls < 20: continue
    per = float(r["TotalDurationNs"]) / 20 / 1e3
    tot += per
    print(f"  {n[:44]:44s} calls/prepare {calls / 20:5.1f}  avg {float(r['AverageNs']) / 1e3:8.2f} us  per prepare {per:8.2f} us")
print(f"  kernels per prepare: {tot:.1f} us")
PY
done
cat $OUT/r06_prepare_kernels.txt
