#!/bin/bash
# round 5: factor-loop stamps and ablations of the persistent solve (tools/ldlt builds).  usage (GPU box): bash tools/dev/r05_fl.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/${1:-r05_fl}; mkdir -p $O
cd $R/tools/ldlt
for b in ldlt_bench_stamps ldlt_bench_abl*; do [ -x $b ] && { echo "== $b"; timeout 60 ./$b 49 2>&1 | head -${2:-60}; }; done > $O/fl.txt 2>&1
cat $O/fl.txt
