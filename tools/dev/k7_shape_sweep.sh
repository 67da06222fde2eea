#!/bin/bash
# (the switches below exist in the measurement build only: make -C ptam_cg_amd/csrc ab)
export PTAM_HIP_LIB=${PTAM_HIP_LIB:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/tools/_ab/libptam_hip.so}
# K7 launch shapes at the headline size: one chunk per wave (product) against the looping form with fewer, longer waves
# (round 2b, coordinate-major W: every looping shape lands at 11.7-13.1 us warm against 11.6 for the product shape)
cd ${GRAFT_REPO_ROOT:-.}
python tools/k7_only.py 50 5000 200 0 19 2>&1 | grep K7
for t in 256 512 1024; do for w in 1 2 3 4; do
  echo -n "loop threads=$t wg/cu=$w: "; PTAM_K7_LOOP=1 PTAM_K7_THREADS=$t PTAM_K7_WG_PER_CU=$w python tools/k7_only.py 50 5000 200 0 19 2>&1 | grep K7
done; done
python tools/k7_only.py 50 5000 200 0 19 2>&1 | grep K7
