# round-end evidence: GPU test suite, default bench line, rocprofv3 kernel stats + PMC passes (tools/profile_r02.sh), serving bench
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests -q -m gpu > $O/r02_pytest_gpu_final.log 2>&1; grep "passed\|failed" $O/r02_pytest_gpu_final.log | tail -2
timeout 600 python bench.py > $O/r02_bench_final.json 2> $O/r02_bench_final.err; head -3 $O/r02_bench_final.err
bash tools/profile_r02.sh > $O/profile_r02.log 2>&1; tail -3 $O/profile_r02.log
STREAM=1 SECONDS=3 timeout 300 python tools/serve_bench.py > $O/r02_serve_bench.log 2>&1; tail -8 $O/r02_serve_bench.log
