#!/bin/bash
# A/B of the camera solve's forms (tools/ldlt/ldlt_bench: us per solve incl. a ~5 us device copy of the system), then the same
# switch on the bench's two bundle legs.   usage: bash tools/ldlt/ab.sh   (from the repository root, on a GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R/tools/ldlt
echo "shape (free cameras [band in block rows]) | default form | PTAM_LDLT_NO_CHAIN=1 (launch per block column / per twin step)"
for a in "14" "20" "33" "49" "60" "69" "80 5" "100 3" "120 2" "128 8" "149 9" "200 4"; do
  d=$(./ldlt_bench $a | head -1 | sed 's/.*: \([0-9.]*\) us per solve.*/\1/')
  n=$(PTAM_LDLT_NO_CHAIN=1 ./ldlt_bench $a | head -1 | sed 's/.*: \([0-9.]*\) us per solve.*/\1/')
  printf "%-10s %8s %8s\n" "$a" "$d" "$n"
done
echo "two-ended shapes: two persistent chains | PTAM_LDLT_TWIN_LAUNCHES=1 (twin-step launches + persistent middle)"
for a in "100 3" "120 2" "160 4" "181 3" "200 4"; do
  d=$(./ldlt_bench $a | head -1 | sed 's/.*: \([0-9.]*\) us per solve.*/\1/')
  n=$(PTAM_LDLT_TWIN_LAUNCHES=1 ./ldlt_bench $a | head -1 | sed 's/.*: \([0-9.]*\) us per solve.*/\1/')
  printf "%-10s %8s %8s\n" "$a" "$d" "$n"
done
cd $R
for v in default PTAM_LDLT_NO_CHAIN; do
  if [ $v = default ]; then unset PTAM_LDLT_NO_CHAIN; else export PTAM_LDLT_NO_CHAIN=1; fi
  python bench.py --no-tracking --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); g=j['global_ba_single_gpu']; l=j['local_ba_config4']
print('$v: headline %.0f trials/s, accepted trial %.1f us, solve %.1f us | config 5: %.0f trials/s, solve %.1f us | config 4: %.0f trials/s, solve %.1f us' % (j['value'], j['accepted_trial_us'], 1e3*j['kernel_ms_per_trial']['solve'], g['value'], 1e3*g['kernel_ms_per_trial']['solve'], l['value'], 1e3*l['kernel_ms_per_trial']['solve']))"
done
