python -m pytest tests/test_gpu_parity.py -x -q -k "bundle" 2>&1 | tail -2
# (the switches below exist in the measurement build only: make -C ptam_cg_amd/csrc ab)
export PTAM_HIP_LIB=${PTAM_HIP_LIB:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/tools/_ab/libptam_hip.so}
python -m pytest tests/test_gpu_dist.py -x -q -k "config5" 2>&1 | tail -1
for m in 1 0; do
  if [ $m = 1 ]; then export PTAM_LDLT_ONE_ENDED=1; else unset PTAM_LDLT_ONE_ENDED; fi
  python bench.py --no-tracking --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['global_ba_single_gpu']
print('one_ended=$m', 'headline value %.0f solve %.1f us | config5 value %.0f ms/step %.4f solve %.1f us' % (d['value'], d['kernel_ms_per_trial']['solve']*1e3, g['value'], g['ms_per_step'], g['kernel_ms_per_trial']['solve']*1e3))"
done
