#!/bin/bash
# round 5: pose-loop phases (timing build) + the pose parity tests + the moving sequence, one GPU call.  usage: bash tools/dev/r05_pose.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for n in 1000 60; do
  PTAM_HIP_LIB=$R/tools/_timing/libptam_hip.so python tools/dev/pose_phases.py $n $([ $n = 60 ] && echo coarse) 2>&1 | grep -v amdgpu.ids
done
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pose or calc_pose" 2>&1 | tail -2
python tools/dev/track_seq.py 8 2>&1 | grep -v amdgpu.ids | tail -2
