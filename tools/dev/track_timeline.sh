#!/bin/bash
# kernel timeline of ONE tracked frame (tools/dev/trackmap_only.py, one call per frame): start offset, duration, gap to the
# previous kernel.  usage: tools/dev/track_timeline.sh [frames to print]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o tt -- python $R/tools/dev/trackmap_only.py 200 frame > /tmp/tt.log 2>&1
python3 - "$@" <<PY
import csv, glob, sys
f = glob.glob("/tmp/tt/**/tt_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "pyr_pvs" in r[2] or r[2].startswith("tm_pyr")]
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
first = idx[len(idx) - 20]
last = idx[len(idx) - 20 + nf]
t0 = rows[first][0]; pe = None
for s, e, n in rows[first:last]:
    gap = (s - pe) / 1e3 if pe else 0.0
    pe = e
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {n}")
PY
tail -1 /tmp/tt.log
