#!/bin/bash
# round 5: spread of the local-BA leg (BASELINE configs[3]) over repeated runs on one box.  usage: bash tools/dev/r05_local_spread.sh [runs]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for i in $(seq ${1:-6}); do
  python bench.py --no-global --no-tracking --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); l=j['local_ba_config4']
k=l['kernel_ms_per_trial']
print('run $i: local BA %.1f us per trial mix %s (accepted trial %.1f us), kernels %s | headline %.0f it/s mix %s, accepted trial %.1f us' % (1e3*l['ms_per_step'], l.get('trial_mix'), l.get('accepted_trial_us',0), {a: round(1e3*b,1) for a,b in k.items()}, j['value'], j['trial_mix'], j['accepted_trial_us']))"
done
