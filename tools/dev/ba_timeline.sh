#!/bin/bash
# kernel timeline (all queues) of a few lambda trials of the bench problem: start offset, duration, gap to the previous kernel
# of the same queue.  usage: tools/dev/ba_timeline.sh [first kernel index] [count]   (environment passes through;
# BA_TIMELINE_ARGS="--cams 20 --points 3000" for another shape)
R=${GRAFT_REPO_ROOT:-.}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -o bt -- python $R/bench.py --no-cpu-baseline --no-tracking --no-global --no-local --steps 10 --jac-reps 1 $BA_TIMELINE_ARGS > /tmp/bt.log 2>&1
python3 - "$@" <<PY
import csv, glob, sys
f = glob.glob("/tmp/bt/**/bt_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][:34]) for r in csv.DictReader(open(f))]
rows.sort()
# the last Compute() with trials: find the last 12 schur_reduce kernels on the busiest queue and print around them
# a window of the timed Compute(): with the second queue active, start 25 kernels before the 7th kernel of the rarer queue;
# otherwise (PTAM_NO_REJECT_SPECULATION=1) 8 trials before the end of the run's first half
import collections
qs = collections.Counter(r[2] for r in rows)
if len(qs) > 1 and "NOSPEC" not in sys.argv:
    rare = min(qs, key=qs.get)
    ii = [i for i, r in enumerate(rows) if r[2] == rare]
    first = max(0, ii[min(6, len(ii) - 1)] - 25)
else:
    idx = [i for i, r in enumerate(rows) if r[3].startswith("finalize_new")]
    first = idx[len(idx) // 2 - 4]
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 75
t0 = rows[first][0]; last_end = {}
for s, e, q, n in rows[first:first + cnt]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"q{q} {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {n}")
PY
