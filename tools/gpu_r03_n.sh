cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_lab_ab.py tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -5
timeout 300 python bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic > $O/r03_n.json 2> $O/r03_n.err
echo "product: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_n.json) $(grep -o '"ms_median": [0-9.]*' $O/r03_n.json)"; grep "enc\.\|layernorm\|stack\|total" $O/r03_n.err
