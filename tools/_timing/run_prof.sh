cd $GRAFT_REPO_ROOT
python tools/k7_only.py 50 5000 100; python tools/k7_only.py 200 50000 50 16
mkdir -p gpurun_out/prof_e && R=$GRAFT_REPO_ROOT && cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e -o k7 -- python $R/tools/k7_only.py 50 5000 100 > $R/gpurun_out/prof_e/log.txt 2>&1; grep -h "jac_accum" $R/gpurun_out/prof_e/*stats*.csv | head -3
