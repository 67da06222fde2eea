#!/bin/bash
# round 5: cycle stamps of the Schur tile kernel's group loop (tools/_exp/schur_stamps = build_variant.sh schur_stamps "-DSCHUR_STAMPS")
# usage (GPU box): bash tools/dev/r05_schur_stamps.sh [variant]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
V=${1:-schur_stamps}
O=$R/gpurun_out/r05_$V; mkdir -p $O
cd $R
PTAM_HIP_LIB=$R/tools/_exp/$V/libptam_hip.so timeout 300 python bench.py --no-cpu-baseline --no-tracking --no-global --no-local --steps 6 --warmup 1 --jac-reps 5 > $O/log.txt 2>&1
L=$(grep -n "SCHUR workgroups" $O/log.txt | tail -2 | head -1 | cut -d: -f1)
tail -n +$L $O/log.txt | head -170 | cut -c1-1200 > $O/stamps.txt
python3 $R/tools/dev/schur_wg_timeline.py $O/stamps.txt v
