#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  tools/collect_profiles.sh <tag>
#   gpurun_out/<tag>/bench.json            python bench.py (default flags)
#   gpurun_out/<tag>/kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/pmc_<tag>_headline/...      PMC passes of the roofline kernel (tools/profile_k7.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
for attempt in 1 2; do   # (rocprofv3 segfaulted once in a tool thread while tracing this command; the second attempt passed)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ba -- python $R/bench.py --no-cpu-baseline --no-replicas > $OUT/trace.log 2>&1 && break
done
cp $OUT/trace/ba_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
bash $R/tools/profile_k7.sh ${TAG}_headline 50 5000 20 > $OUT/pmc_headline.txt 2>&1
bash $R/tools/profile_k7.sh ${TAG}_config5 200 50000 20 16 > $OUT/pmc_config5.txt 2>&1
# PMC passes of the Schur tile kernel and the LDL^T step (MFMA / VALU / LDS / wait counters) over a short BA run
for pass in "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" \
            "sq2 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_ba -o $name -- python $R/bench.py --no-cpu-baseline --no-tracking --no-global --steps 10 --warmup 1 > $OUT/pmc_ba_$name.log 2>&1 || echo "pass $name failed"
done
python3 - <<PY | tee $OUT/pmc_schur_ldlt.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_ba/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(x in k for x in ("schur_tile", "ldlt_step", "ldlt_backward", "ldlt_chain", "ldlt_small")): continue
        a = agg[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        print(f"{k[:28]:28s} {c:26s} per-launch {v / max(n,1):.4g}  (launches {n})")
PY
# the keyed traffic record bench.py reads (profiles/k7_pmc_traffic.json): written next to the summaries, copy it into profiles/
python3 - <<PY
import json
rec = {}
for key, dd in (("bundle_50kf_x_5000pts_dense", "$R/gpurun_out/pmc_${TAG}_headline"), ("bundle_200kf_x_50000pts_window16", "$R/gpurun_out/pmc_${TAG}_config5")):
    try:
        rec[key] = json.load(open(dd + "/traffic.json"))
    except OSError:
        pass
import sys
sys.path.insert(0, "$R")
from ptam_cg_amd._srchash import source_sha16
rec["source_sha16"] = source_sha16()
json.dump(rec, open("$OUT/k7_pmc_traffic.json", "w"), indent=1)
PY
# the tracked frame, kernel by kernel: rocprofv3 stats of tools/dev/track_seq.py (the moving-camera sequence bench.py tracks: one
# ptam_track_frame per frame), per-frame launch counts and average durations -> tracking_kernels.json (bench.py embeds the committed
# copy in `tracking` when its source fingerprint is the running library's)
cd /tmp
PASSES=4
FR=$((64 * (2 + 3 * PASSES) / 2))
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ttrace -o tm -- python $R/tools/dev/track_seq.py $PASSES > $OUT/tracking_trace.log 2>&1
cp $OUT/ttrace/tm_kernel_stats.csv $OUT/tracking_kernel_stats.csv 2>/dev/null
python3 - <<PY
import csv, json
frames = 2 * $FR   # (trackmap_only.py runs the loop twice)
import sys
sys.path.insert(0, "$R")
from ptam_cg_amd._srchash import source_sha16
rec = {"source": "rocprofv3 --kernel-trace --stats -- python tools/dev/track_seq.py $PASSES (the moving-camera sequence)", "frames": frames,
       "source_sha16": source_sha16(), "kernels": {}}
tot = 0.0
try:
    for r in csv.DictReader(open("$OUT/tracking_kernel_stats.csv")):
        n = r["Name"].split("(")[0]
        if "rocclr" in n or int(r["Calls"]) < frames // 2: continue   # (set-up kernels: keyframe of the map, uploads)
        per = int(r["Calls"]) / frames
        rec["kernels"][n] = {"launches_per_frame": round(per, 2), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                             "us_per_frame": round(per * float(r["AverageNs"]) / 1e3, 2)}
        tot += per * float(r["AverageNs"]) / 1e3
    rec["kernel_us_per_frame"] = round(tot, 1)
except OSError as e:
    rec["error"] = repr(e)
json.dump(rec, open("$OUT/tracking_kernels.json", "w"), indent=1)
print(json.dumps(rec)[:600])
PY
rm -rf $OUT/ttrace
cd $R
# gpurun merges at most 64 MiB back: keep the summaries, drop the raw per-dispatch tables
rm -rf $OUT/pmc_ba $OUT/trace
for dd in $R/gpurun_out/pmc_${TAG}_headline $R/gpurun_out/pmc_${TAG}_config5; do
  find $dd -name '*_kernel_trace.csv' -delete; find $dd -name '*_counter_collection.csv' -delete; find $dd -name '*_agent_info.csv' -delete
done
tail -3 $OUT/pmc_headline.txt; head -c 600 $OUT/bench.json
