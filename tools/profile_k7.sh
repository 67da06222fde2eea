#!/bin/bash
# rocprofv3 PMC passes for the roofline kernel (one counter group per pass, as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).  Usage: tools/profile_k7.sh <tag> [args of k7_only.py]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-k7}; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- python $R/tools/k7_only.py $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
ARGS="$*"
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections, json
vals = {}
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "jac_accum" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items():
        vals[k] = v / max(n, 1)
        print(f"{f.split('/')[-1][:12]:12s} {k:24s} per-launch {v / max(n,1):.4g}  (launches {n})")
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM)
    rd = 2.0 * vals["FETCH_SIZE"] * 1024.0
    wr = vals["WRITE_SIZE"] * 1024.0
    print(f"HBM traffic per launch: read {rd/1e6:.2f} MB (FETCH_SIZE x2) + write {wr/1e6:.2f} MB = {(rd+wr)/1e6:.2f} MB")
    json.dump({"fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"], "read_bytes": rd, "write_bytes": wr,
               "traffic_bytes_per_launch": rd + wr}, open("$OUT/traffic.json", "w"))
PY
