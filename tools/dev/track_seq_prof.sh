#!/bin/bash
# rocprofv3 kernel stats of the moving-camera sequence + the pose kernel's phase stamps (timing build):  tools/dev/track_seq_prof.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-seq}
OUT=$R/gpurun_out/seq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/dev/track_seq.py 8 stages 2>&1 | grep -v amdgpu.ids | tee $OUT/plain.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o seq -- python $R/tools/dev/track_seq.py 4 > $OUT/log.txt 2>&1
python3 - <<PY | tee $OUT/kernels.txt
import csv
rows = list(csv.DictReader(open("$OUT/seq_kernel_stats.csv")))
nf = None
for r in rows:
    if "tm_compact_select" in r["Name"]:
        nf = int(r["Calls"])
tot = 0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    c = int(r["Calls"])
    if nf and c >= nf // 2:
        per = float(r["TotalDurationNs"]) / nf / 1e3
        tot += per
        print(f'{r["Name"].split("(")[0][:52]:52s} calls/frame {c / nf:4.1f} avg {float(r["AverageNs"]) / 1e3:7.2f} us  per frame {per:7.2f}')
print("kernel us per frame", round(tot, 1), "frames", nf)
PY
if [ -f $R/tools/_timing/libptam_hip.so ]; then
  for n in 1000 60; do
    PTAM_HIP_LIB=$R/tools/_timing/libptam_hip.so python $R/tools/dev/pose_phases.py $n $([ $n = 60 ] && echo coarse) 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phases.txt
  done
fi
