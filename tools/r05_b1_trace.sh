cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/b1_trace.py
rm -rf $R/gpurun_out/prof_b1; timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b1 -- python $R/tools/b1_trace.py > $R/gpurun_out/r05_b1_trace.log 2>&1; tail -2 $R/gpurun_out/r05_b1_trace.log
cd $R; python tools/rocprof_summary.py stats gpurun_out/prof_b1 > gpurun_out/r05_b1_rocprof_stats.txt 2>&1; head -40 gpurun_out/r05_b1_rocprof_stats.txt | cut -c1-140; rm -rf gpurun_out/prof_b1
