#!/bin/bash
# tools/ldlt/prof.sh BINARY ARGS...: rocprofv3 per-kernel averages of one ldlt_bench run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
B=$1; shift
OUT=$R/gpurun_out/ldlt_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- $R/tools/ldlt/$B "$@" 2>/dev/null | grep "per solve"
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/p_kernel_stats.csv")):
    print(f'  {r["Name"].split("(")[0][:40]:40s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:7.2f} us  min {float(r["MinNs"])/1e3:6.2f} max {float(r["MaxNs"])/1e3:6.2f}')
PY
