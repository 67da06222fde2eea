#!/bin/bash
# (the switches below exist in the measurement build only: make -C ptam_cg_amd/csrc ab)
export PTAM_HIP_LIB=${PTAM_HIP_LIB:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/tools/_ab/libptam_hip.so}
# A/B of the Schur tile kernel: settings of the work split (env) and experimental builds under tools/_exp/ (tools/dev/build_variant.sh), one GPU call
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $1"; shift; env "$@" bash tools/dev/kstats.sh ab "schur_tile" | sed 's/^/   /'; }
for rep in 1 2; do
run "product" A=1
run "no pattern sort" PTAM_SCHUR_SORT=0
for v in $(ls tools/_exp 2>/dev/null); do
  [ -f tools/_exp/$v/libptam_hip.so ] && run "$v" PTAM_HIP_LIB=$PWD/tools/_exp/$v/libptam_hip.so
done
done
