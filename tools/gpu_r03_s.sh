cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_session.py tests/test_gpu_parity.py -q -x -m gpu -k "session or b1 or injected or real_hidden_width or golden or vctk_low_ragged" 2>&1 | tail -3
timeout 300 python bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic > $O/r03_s.json 2> $O/r03_s.err
echo "product: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_s.json) $(grep -o '"ms_median": [0-9.]*' $O/r03_s.json)"; grep "total" $O/r03_s.err
