#!/bin/bash
# rocprofv3 kernel stats of batched tracking: tools/dev/batch_stats.sh [frames per batch]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
K=${1:-64}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bs
BATCH=1 NF=50 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs -o bs -- python $R/tools/dev/replicas.py $K 2>&1 | grep -E "contexts|batched"
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/bs/**/bs_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    if "batch" in n:
        print(f'{n[:48]:48s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f}')
PY
