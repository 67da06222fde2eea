#!/bin/bash
# rocprofv3 kernel stats of the resident TrackMap chain (tools/dev/trackmap_only.py):  tools/dev/tstats.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-ts}
OUT=$R/gpurun_out/ts_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tm -- python $R/tools/dev/trackmap_only.py ${FRAMES:-300} ${MODE:-frame} > $OUT/log.txt 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/tm_kernel_stats.csv")):
    n = r["Name"].split("(")[0]
    if "rocclr" in n: continue
    print(f'{n[:44]:44s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f}')
PY
grep -E "frame|us" $OUT/log.txt | tail -3
find $OUT -name '*_kernel_trace.csv' -delete
