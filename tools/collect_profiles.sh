#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  tools/collect_profiles.sh <tag>
#   gpurun_out/<tag>/bench.json            python bench.py (default flags)
#   gpurun_out/<tag>/kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/pmc_<tag>_headline/...      PMC passes of the roofline kernel (tools/profile_k7.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ba -- python $R/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
cp $OUT/trace/ba_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
bash $R/tools/profile_k7.sh ${TAG}_headline 50 5000 20 > $OUT/pmc_headline.txt 2>&1
bash $R/tools/profile_k7.sh ${TAG}_config5 200 50000 20 16 > $OUT/pmc_config5.txt 2>&1
tail -3 $OUT/pmc_headline.txt; head -c 600 $OUT/bench.json
