#!/bin/bash
# round 5: rocprofv3 kernel stats of the headline bundle leg for the product and every build under tools/_exp, filtered by a regex
# usage (GPU box): bash tools/dev/r05_kstat_variants.sh <regex>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/ks_$1; mkdir -p /tmp/ks_$1
  PTAM_HIP_LIB=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$1 -o ba -- python $R/bench.py --no-cpu-baseline --no-tracking --no-global --no-local > /tmp/ks_$1/log.txt 2>&1
  python3 - "$1" "$3" <<PY
import csv, re, sys
for r in csv.DictReader(open("/tmp/ks_%s/ba_kernel_stats.csv" % sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f'{sys.argv[1]:14s} {r["Name"].split("(")[0][:34]:34s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f} max {float(r["MaxNs"])/1e3:7.2f}')
PY
}
run product $R/ptam_cg_amd/csrc/libptam_hip.so "${1:-.}"
for v in $(ls $R/tools/_exp 2>/dev/null); do [ -f $R/tools/_exp/$v/libptam_hip.so ] && run $v $R/tools/_exp/$v/libptam_hip.so "${1:-.}"; done
