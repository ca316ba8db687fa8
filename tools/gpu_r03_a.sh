# round 3, first GPU pass: suite + bench with per-kernel table, A/B of k_mrf_p against the split-per-tap kernel
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x > $O/r03_pytest_gpu_a.log 2>&1; tail -5 $O/r03_pytest_gpu_a.log
timeout 600 python bench.py --steps 50 --no-extra --no-cpu-baseline > $O/r03_bench_a.json 2> $O/r03_bench_a.err; cat $O/r03_bench_a.json | cut -c1-600; grep -i "mrf\|wn\|ms" $O/r03_bench_a.err | head -40
MI355VITS_NO_MRF_P=1 timeout 600 python bench.py --steps 50 --no-extra --no-cpu-baseline --no-b1 > $O/r03_bench_a_nop.json 2> $O/r03_bench_a_nop.err; cat $O/r03_bench_a_nop.json | cut -c1-300; grep -i "mrf" $O/r03_bench_a_nop.err | head
