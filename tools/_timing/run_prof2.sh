cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_f && R=$GRAFT_REPO_ROOT && cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f -o ba -- python $R/bench.py --no-tracking --no-cpu-baseline > $R/gpurun_out/prof_f/log.txt 2>&1; grep -h "ldlt\|schur\|jac_accum" $R/gpurun_out/prof_f/*kernel_stats.csv | cut -c1-60,60-200 | head
cd $R && PTAM_HIP_LIB=tools/_timing/libptam_hip.so python bench.py --no-tracking --no-cpu-baseline 2>&1 | grep -E "LDLT|SCHUR"
