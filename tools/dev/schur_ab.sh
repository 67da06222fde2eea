#!/bin/bash
# A/B of the Schur tile kernel: work-split settings (and ablation builds under tools/_exp/, if present), one GPU call
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $1"; shift; env "$@" bash tools/dev/kstats.sh ab "schur_tile" | sed 's/^/   /'; }
run "default" A=1
run "nx8" PTAM_SCHUR_NX=8
run "nx8 cost=frags+3" PTAM_SCHUR_NX=8 PTAM_SCHUR_COST=3
run "nx8 cost=frags+8" PTAM_SCHUR_NX=8 PTAM_SCHUR_COST=8
run "default again" A=1
run "nx8 again" PTAM_SCHUR_NX=8
for v in nomfma noload; do
  [ -f tools/_exp/$v/libptam_hip.so ] && run "$v" PTAM_HIP_LIB=$PWD/tools/_exp/$v/libptam_hip.so
done
