#!/bin/bash
# rocprofv3 kernel stats of a short BA bench, filtered:  tools/dev/kstats.sh <tag> [regex]   (env passes through)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-ks}; RE=${2:-.}
OUT=$R/gpurun_out/ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ba -- python $R/bench.py --no-cpu-baseline --no-tracking > $OUT/log.txt 2>&1
python3 - <<PY
import csv, re
for r in csv.DictReader(open("$OUT/ba_kernel_stats.csv")):
    if re.search(r"$RE", r["Name"]):
        print(f'{r["Name"].split("(")[0][:44]:44s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f} max {float(r["MaxNs"])/1e3:7.2f}')
PY
python3 - <<PY
import json
b = json.loads([l for l in open("$OUT/log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {})
print("headline: %.0f trials/s, schur %.1f us, K7 %.2f / %.2f cold | config 5: %.0f trials/s, schur %.1f us, jacobian %.1f us" % (
    b["value"], 1e3 * b["kernel_ms_per_trial"]["schur"], b["roofline"]["avg_launch_us"], b["roofline"].get("avg_launch_us_cold", 0),
    g.get("value", 0), 1e3 * g.get("kernel_ms_per_trial", {}).get("schur", 0), 1e3 * g.get("kernel_ms_per_trial", {}).get("jacobian", 0)))
PY
