cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_t && R=$GRAFT_REPO_ROOT && cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_t -o tr -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_t/log.txt 2>&1
grep -h "pose_gn\|zmssd_search\|fast_\|pyramid" $R/gpurun_out/prof_t/tr_kernel_stats.csv | cut -d, -f1-4 | cut -c1-40,100-200
