cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > $O/r05_pytest_gpu_head.log 2>&1; grep "passed\|failed" $O/r05_pytest_gpu_head.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_head_driver_args.json 2> $O/r05_bench_head_driver_args.err; grep -o '"ms_per_step": [0-9.]*' $O/r05_bench_head_driver_args.json | head -1
