cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_lab_ab.py -q -x -m gpu 2>&1 | tail -3
timeout 300 python bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic > $O/r03_p.json 2> $O/r03_p.err
echo "product: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_p.json) $(grep -o '"ms_median": [0-9.]*' $O/r03_p.json)"; grep "enc\.\|layernorm\|total" $O/r03_p.err
