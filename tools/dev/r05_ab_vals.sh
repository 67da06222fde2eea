R=${GRAFT_REPO_ROOT}
cd $R
for rep in 1 2 3 4 5 6; do
for lib in $R/ptam_cg_amd/csrc/libptam_hip.so $R/tools/_exp/head/libptam_hip.so; do
  PTAM_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-tracking > /tmp/ab_log.txt 2>&1
  python3 - "$lib" <<PY
import json, sys
b = json.loads([l for l in open("/tmp/ab_log.txt") if l.startswith("{")][-1])
g = b.get("global_ba_single_gpu", {}); l = b.get("local_ba_config4", {})
print("%-6s accepted %.1f us | config5 %.0f it/s (%.1f us) | local %.1f us | mix %s" % (sys.argv[1].split("/")[-2], b.get("accepted_trial_us", 0), g.get("value", 0), 1e3*g.get("ms_per_step",0), 1e3 * l.get("ms_per_step", 0), list(b.get("trial_mix", {}).values())))
PY
done
done
