cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_lab_ab.py tests/test_gpu_parity.py -q -x -m gpu -k "dds_stack or real_hidden_width or b1 or golden or vctk_low_ragged" > $O/r03_t_pytest.log 2>&1; grep -E "passed|failed|error" $O/r03_t_pytest.log | tail -3
timeout 300 python bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic > $O/r03_t.json 2> $O/r03_t.err
echo "product: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_t.json) $(grep -o '"ms_median": [0-9.]*' $O/r03_t.json)"; grep "stack\|total" $O/r03_t.err
