#!/bin/bash
# (the switches below exist in the measurement build only: make -C ptam_cg_amd/csrc ab)
export PTAM_HIP_LIB=${PTAM_HIP_LIB:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/tools/_ab/libptam_hip.so}
# A/B of an environment switch on the frame chain:  tools/dev/ab_env.sh VAR   (native one-context driver, alternating runs)
V=$1
for i in 1 2 3; do
  echo -n "default:  "; NF=3000 timeout 100 python tools/dev/replicas.py 1 2>&1 | grep contexts
  echo -n "$V=1: "; env $V=1 NF=3000 timeout 100 python tools/dev/replicas.py 1 2>&1 | grep contexts
done
