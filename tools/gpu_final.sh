# round-end evidence: GPU test suite, default bench line (with the extra legs), the math-mode error table vs the fp64 oracle;
# profiles (rocprofv3 stats + PMC) come from tools/profile_r02.sh
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests -q -m gpu > $O/r02_pytest_gpu_final.log 2>&1; grep "passed\|failed" $O/r02_pytest_gpu_final.log | tail -2
timeout 600 python bench.py > $O/r02_bench_final.json 2> $O/r02_bench_final.err; head -3 $O/r02_bench_final.err
timeout 300 python tools/math_modes_vs_fp64.py > $O/r02_math_modes_vs_fp64.log 2>/dev/null; cat $O/r02_math_modes_vs_fp64.log
timeout 300 python bench.py --math f16x2 --no-extra --steps 100 > $O/r02_bench_f16x2.json 2> $O/r02_bench_f16x2.err; head -3 $O/r02_bench_f16x2.err
