cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "attention or parity or golden or batch" 2>&1 | tail -3
timeout 300 python bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic > $O/r03_o.json 2> $O/r03_o.err
echo "product: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_o.json) $(grep -o '"ms_median": [0-9.]*' $O/r03_o.json)"; grep "enc\.\|layernorm\|stack\|total" $O/r03_o.err
