# one gpurun call: the kernel tests touched this round, a bench line with the per-kernel table, rocprofv3 kernel stats
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_transpose1d or bench_workload or split" > $O/q_pytest.log 2>&1; tail -3 $O/q_pytest.log
timeout 400 python bench.py --steps 100 > $O/q_bench.json 2> $O/q_bench.err; tail -c 600 $O/q_bench.json; grep -A28 "^per-kernel" $O/q_bench.err | head -40
MI355VITS_NO_B3_PC=1 timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-extra --no-b1 > $O/q_bench_nopc.json 2> $O/q_bench_nopc.err; grep -A12 "^per-kernel" $O/q_bench_nopc.err | grep "upsample\|mrf\|wn_layer"; head -2 $O/q_bench_nopc.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/prof_stats
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -- python $R/bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-extra --no-b1 --no-roofline > $R/$O/rocprof_stats.log 2>&1
cd $R
python tools/rocprof_summary.py stats $O/prof_stats > $O/q_rocprof_stats.txt 2>&1; head -16 $O/q_rocprof_stats.txt | cut -c1-150
rm -rf $O/prof_stats
