for rep in 1 2; do
for v in product prio1 prio2 prio3 flush prio1flush; do
  if [ $v = product ]; then unset PTAM_HIP_LIB; else export PTAM_HIP_LIB=$PWD/tools/_exp/$v/libptam_hip.so; fi
  python tools/k7_only.py 50 5000 200 0 19 2>&1 | grep K7
done; done
unset PTAM_HIP_LIB
for v in product prio1flush; do
  if [ $v = product ]; then unset PTAM_HIP_LIB; else export PTAM_HIP_LIB=$PWD/tools/_exp/$v/libptam_hip.so; fi
  python tools/k7_only.py 200 50000 50 16 6 2>&1 | grep K7
done
unset PTAM_HIP_LIB
python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -3
