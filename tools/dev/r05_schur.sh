#!/bin/bash
# round 5: the Schur tile kernel at the HEADLINE only — product and the ablation builds under tools/_exp (build_variant.sh), rocprofv3
# kernel stats, then PMC passes of the product.  usage (GPU box): bash tools/dev/r05_schur.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_schur; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, lib
  rm -rf /tmp/ks_$1; mkdir -p /tmp/ks_$1
  PTAM_HIP_LIB=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$1 -o ba -- python $R/bench.py --no-cpu-baseline --no-tracking --no-global --no-local > /tmp/ks_$1/log.txt 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("/tmp/ks_$1/ba_kernel_stats.csv")):
    if "schur" in r["Name"]:
        print(f'$1: {r["Name"].split("(")[0][:30]:30s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f} max {float(r["MaxNs"])/1e3:7.2f}')
PY
}
run product $R/ptam_cg_amd/csrc/libptam_hip.so
for v in $(ls $R/tools/_exp 2>/dev/null); do [ -f $R/tools/_exp/$v/libptam_hip.so ] && run $v $R/tools/_exp/$v/libptam_hip.so; done
for grp in ${SCHUR_PMC-"SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum"}; do
  rm -rf /tmp/pmc; mkdir -p /tmp/pmc
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/bench.py --no-cpu-baseline --no-tracking --no-global --no-local > /tmp/pmc/log.txt 2>&1
  python3 - <<PY
import csv, collections, glob
f = glob.glob("/tmp/pmc/*counter_collection.csv")
acc = collections.defaultdict(lambda: [0.0, 0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "schur_tile" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print(f"pmc schur_tile {k:32s} per-launch {s / n:.4g} (launches {n})")
PY
done
