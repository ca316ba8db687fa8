# round-end evidence: GPU test suite, default bench line, rocprofv3 kernel stats + PMC passes (tools/profile_r02.sh)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests -q -m gpu > $O/r02_pytest_gpu_final.log 2>&1; tail -3 $O/r02_pytest_gpu_final.log
timeout 600 python bench.py > $O/r02_bench_final.json 2> $O/r02_bench_final.err; tail -c 400 $O/r02_bench_final.json; head -3 $O/r02_bench_final.err
bash tools/profile_r02.sh > $O/profile_r02.log 2>&1; tail -5 $O/profile_r02.log
