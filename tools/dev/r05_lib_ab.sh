#!/bin/bash
# round 5: A/B of library builds (product + every tools/_exp/*/libptam_hip.so) on the bench's three bundle legs, two repetitions,
# one GPU call.  usage (GPU box): bash tools/dev/r05_lib_ab.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_lib_ab; mkdir -p $O
cd $R
for rep in 1 2; do
for lib in $R/ptam_cg_amd/csrc/libptam_hip.so $(ls $R/tools/_exp/*/libptam_hip.so 2>/dev/null); do
  PTAM_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-tracking > $O/log.txt 2>&1
  python3 - "$lib" <<PY | tee -a $O/out.txt
import json, sys
b = json.loads([l for l in open("$O/log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
k = lambda d, n: 1e3 * d.get("kernel_ms_per_trial", {}).get(n, 0)
name = sys.argv[1].split("/")[-2]
print("%-14s schur %.1f %.1f %.1f | solve %.1f %.1f %.1f | jac %.1f %.1f %.1f | it/s %.0f %.0f %.0f" % (name, k(b, "schur"), k(g, "schur"), k(l, "schur"),
      k(b, "solve"), k(g, "solve"), k(l, "solve"), k(b, "jacobian"), k(g, "jacobian"), k(l, "jacobian"), b["value"], g.get("value", 0), l.get("value", 0)))
PY
done
done
