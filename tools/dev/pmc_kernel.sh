#!/bin/bash
# PMC passes (per-launch averages) of the kernels matching a regex over a short headline BA run:
#   tools/dev/pmc_kernel.sh <tag> <kernel regex>      (EXTRA="..." adds bench.py arguments)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-pk}; RE=${2:-schur_tile}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" \
            "sq2 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
            "tc1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- python $R/bench.py --no-cpu-baseline --no-tracking --no-global --steps 10 --warmup 1 $EXTRA > $OUT/$name.log 2>&1 || echo "pass $name failed"
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections, re
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not re.search(r"$RE", k): continue
        a = agg[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        print(f"{k[:28]:28s} {c:30s} per-launch {v / max(n,1):.4g}  (launches {n})")
PY
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_counter_collection.csv' -delete; find $OUT -name '*_agent_info.csv' -delete
