#!/bin/bash
# per-workgroup stamps of the Schur tile kernel on one shape + the second fit.  usage (GPU box): bash tools/dev/r05_schur_shape2.sh <tag> "<bench shape args>" "ENV=.." ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=$1; SH=$2; shift 2
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
i=0
for cfg in "$@"; do
  i=$((i+1))
  env PTAM_HIP_LIB=$R/tools/_exp/schur_stamps_ab/libptam_hip.so $cfg timeout 300 python bench.py $SH --no-cpu-baseline --no-tracking --no-global --no-local --steps 6 --warmup 1 --jac-reps 5 > $O/stamps_$i.txt 2>&1
  echo "=== $cfg" | tee -a $O/fit.txt
  python3 tools/dev/schur_fit2.py $O/stamps_$i.txt 2>&1 | tee -a $O/fit.txt
  grep -o '"schur": [0-9.e-]*' $O/stamps_$i.txt | head -1 | tee -a $O/fit.txt
done
