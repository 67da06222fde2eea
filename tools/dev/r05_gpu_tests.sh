#!/bin/bash
# round 5: the GPU test suite + the default bench line in one GPU call.  usage (GPU box): bash tools/dev/r05_gpu_tests.sh <tag> [pytest args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/${1:-r05_tests}; mkdir -p $O
shift
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python3 - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
k=j.get("kernel_ms_per_trial",{})
print("value %.0f it/s  accepted_trial_us %s  mix %s" % (j["value"], j.get("accepted_trial_us"), j.get("trial_mix")))
print("kernel us/trial:", {a: round(1e3*b,1) for a,b in k.items()})
g=j.get("global_ba_single_gpu",{}); l=j.get("local_ba_config4",{})
print("config5 1gpu %.0f it/s solve %.1f | config4 %.0f it/s (%.1f us) solve %.1f" % (g.get("value",0), 1e3*g.get("kernel_ms_per_trial",{}).get("solve",0), l.get("value",0), 1e3*l.get("ms_per_step",0), 1e3*l.get("kernel_ms_per_trial",{}).get("solve",0)))
print("roofline", j.get("roofline",{}).get("frac"), "tracked_fps", j.get("tracking",{}).get("tracked_fps"))
PY
