cd $GRAFT_REPO_ROOT
for g in 1 2 3 0; do
mkdir -p gpurun_out/prof_h$g && R=$GRAFT_REPO_ROOT && cd /tmp && export TMPDIR=/tmp && P1H=$g rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_h$g -o ba -- python $R/bench.py --no-tracking --no-cpu-baseline > $R/gpurun_out/prof_h$g/log.txt 2>&1
cd $R
done
