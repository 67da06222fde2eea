#!/bin/bash
# K7 launch-shape sweep on one problem: straight-line vs looping form, workgroup width, workgroups per CU.
# usage: tools/dev/k7_shape_sweep.sh [cams pts reps window]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
A="${*:-50 5000 50}"
echo "default:"; python $R/tools/k7_only.py $A | tail -1
for t in 256 512; do for w in 1 2 3 4 6 8; do
  echo -n "loop threads=$t wg/cu<=$w: "; PTAM_K7_LOOP=1 PTAM_K7_THREADS=$t PTAM_K7_WG_PER_CU=$w python $R/tools/k7_only.py $A | tail -1
done; done
