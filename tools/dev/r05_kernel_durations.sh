#!/bin/bash
# round 5: every launch's duration of the kernels matching a regex over the headline bundle leg (rocprofv3 kernel trace), as a sorted list
# usage (GPU box): bash tools/dev/r05_kernel_durations.sh <regex>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kd; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kd -o kd -- python $R/bench.py --no-cpu-baseline --no-tracking --no-global --no-local > /tmp/kd.log 2>&1
python3 - "${1:-.}" <<PY
import csv, glob, re, sys, collections
f = glob.glob("/tmp/kd/**/kd_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if re.search(sys.argv[1], n): d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    v.sort()
    print(n[:40], len(v), "durations us:", " ".join("%.1f" % x for x in v[:: max(1, len(v) // 40)]))
PY
