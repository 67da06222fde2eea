for v in base ${VARIANT:-pose512}; do
  if [ $v = base ]; then unset PTAM_HIP_LIB; else export PTAM_HIP_LIB=tools/_exp/$v/libptam_hip.so; fi
  echo "== $v"
  timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pose" 2>&1 | tail -1
  NF=3000 timeout 100 python tools/dev/replicas.py 1 2>&1 | grep contexts
  NF=3000 timeout 100 python tools/dev/replicas.py 1 2>&1 | grep contexts
done
