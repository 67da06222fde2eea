#!/bin/bash
# kernel timeline of k concurrent tracker contexts: per queue, kernel time and gaps between consecutive kernels
K=${1:-8}
R=${GRAFT_REPO_ROOT:-.}
cd /tmp && export TMPDIR=/tmp
NF=300 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -o rt -- python $R/tools/dev/replicas.py $K 2>&1 | grep contexts
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/rt/**/rt_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
byq = collections.defaultdict(list)
for r in rows:
    byq[(r.get("Queue_Id"), r.get("Stream_Id"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
print("queues/streams:", len(byq))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:6]:
    ks.sort()
    ks = ks[len(ks) // 2:]   # second pass (timed)
    busy = sum(e - s for s, e, _ in ks); span = ks[-1][1] - ks[0][0]
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    gaps.sort()
    print(q, "kernels", len(ks), "busy %.1f%%" % (100.0 * busy / span), "gap median %.1f us p90 %.1f us max %.1f" % (gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * .9)] / 1e3, gaps[-1] / 1e3))
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n in ks: per[n][0] += e - s; per[n][1] += 1
    print("   ", {n: round(v[0] / v[1] / 1e3, 1) for n, v in per.items()})
PY
