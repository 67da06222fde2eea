#!/bin/bash
# round 5: the camera solve's microbench at a few shapes, plain + timing build + rocprofv3 per kernel.  usage (GPU box): bash tools/dev/r05_ldlt_base.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/${1:-r05_ldlt}; mkdir -p $O
cd $R/tools/ldlt
{ for f in 8 14 19 33 49 69; do ./ldlt_bench $f; done; ./ldlt_bench 200 4; ./ldlt_bench 100 3; ./ldlt_bench 128 8; ./ldlt_bench 80 5
  echo "-- PTAM_LDLT_SEPARATE_BACKWARD=1"; for f in 19 33 49 69; do PTAM_LDLT_SEPARATE_BACKWARD=1 ./ldlt_bench $f; done; PTAM_LDLT_SEPARATE_BACKWARD=1 ./ldlt_bench 128 8; } > $O/ldlt_plain.txt 2>&1
[ -x ./ldlt_bench_timing ] && { ./ldlt_bench_timing 49 > $O/ldlt_timing49.txt 2>&1; ./ldlt_bench_timing 19 > $O/ldlt_timing19.txt 2>&1; }
bash prof.sh ldlt_bench 49 > $O/prof49.txt 2>&1
cat $O/ldlt_plain.txt $O/prof49.txt
