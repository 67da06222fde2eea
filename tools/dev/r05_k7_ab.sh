#!/bin/bash
# round 5: K7 alone (tools/k7_only.py: hot loop of 200 launches + the cold round-robin) for the product and every build under tools/_exp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
for lib in $R/ptam_cg_amd/csrc/libptam_hip.so $(ls $R/tools/_exp/*/libptam_hip.so 2>/dev/null); do
  echo "$(basename $(dirname $lib)): $(PTAM_HIP_LIB=$lib python tools/k7_only.py 50 5000 200 0 19 2>&1 | grep K7 | tr '\n' ' ')"
done; done
