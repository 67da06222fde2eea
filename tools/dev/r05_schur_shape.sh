#!/bin/bash
# round 5: per-workgroup stamps of the Schur tile kernel on one shape, for a list of env settings (stamps build with the
# measurement switches: tools/dev/build_variant.sh schur_stamps_ab "-DSCHUR_STAMPS -DPTAM_AB_SWITCHES")
# usage (GPU box): bash tools/dev/r05_schur_shape.sh <tag> <cams> <points> <free cameras> "ENV=.." ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
T=$1; C=$2; P=$3; F=$4; shift 4
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
i=0
for cfg in "$@"; do
  i=$((i+1))
  env PTAM_HIP_LIB=$R/tools/_exp/schur_stamps_ab/libptam_hip.so $cfg timeout 300 python bench.py --cams $C --points $P --no-cpu-baseline --no-tracking --no-global --no-local --steps 6 --warmup 1 --jac-reps 5 > $O/stamps_$i.txt 2>&1
  echo "=== $cfg" | tee -a $O/fit.txt
  python3 tools/dev/schur_fit.py $O/stamps_$i.txt $F 2>&1 | grep -v "^(" | grep -A3 "per CU" | tee -a $O/fit.txt
  python3 tools/dev/schur_fit.py $O/stamps_$i.txt $F 2>&1 | grep "^(\|by XCD" > $O/cus_$i.txt
  grep -o '"schur": [0-9.e-]*' $O/stamps_$i.txt | head -1 | tee -a $O/fit.txt
done
