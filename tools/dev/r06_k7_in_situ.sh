#!/bin/bash
# round 6: K7's duration INSIDE Compute() from the kernel trace (no event bracket): the launches of jac_accum_wave_kernel whose
# predecessor on the queue is select_final_kernel, against the back-to-back launches of the roofline leg.
#   -> gpurun_out/r06_k7_in_situ.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k7_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/k7_trace -o t -- python $R/bench.py --no-cpu-baseline --no-tracking --no-global --no-local > /tmp/k7_trace.log 2>&1
python3 - > $OUT/r06_k7_in_situ.txt <<PY
import csv, glob, statistics
f = glob.glob("/tmp/k7_trace/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.split("(")[0].replace("void ", "")
prev = None
situ, warm, gaps = [], [], []
by_prev = {}
for r in rows:
    n = short(r["Kernel_Name"])
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n.startswith("jac_accum_wave_kernel<1024"):
        p = short(prev["Kernel_Name"]) if prev else "-"
        if p.startswith("select_final"):
            situ.append(dur)
            gaps.append((int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3)
        elif p.startswith("jac_accum_wave_kernel<1024"):
            warm.append(dur)
    prev = r
q = lambda v, p: sorted(v)[int(p * (len(v) - 1))]
print("K7 (jac_accum_wave_kernel<1024,...>, 50 x 5000) durations from the kernel trace, us:")
print(f"  back to back (roofline leg)   n {len(warm):6d}  median {statistics.median(warm):6.2f}  p10 {q(warm, .1):6.2f}  p90 {q(warm, .9):6.2f}")
print(f"  inside Compute() (behind select_final_kernel) n {len(situ):4d}  median {statistics.median(situ):6.2f}  p10 {q(situ, .1):6.2f}  p90 {q(situ, .9):6.2f}")
print(f"  gap between select_final_kernel's end and K7's start inside Compute(): median {statistics.median(gaps):5.2f} us")
PY
cat $OUT/r06_k7_in_situ.txt
