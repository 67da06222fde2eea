python -m pytest tests/test_gpu_parity.py -x -q -k "bundle" 2>&1 | tail -2
for m in block wave; do
  PTAM_LDLT=$m python bench.py --no-tracking --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['global_ba_single_gpu']
print('$m', 'headline value %.0f ms/step %.4f solve %.1f us | config5 value %.0f solve %.1f us' % (d['value'], d['ms_per_step'], d['kernel_ms_per_trial']['solve']*1e3, g['value'], g['kernel_ms_per_trial']['solve']*1e3))"
done
