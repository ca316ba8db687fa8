# DDS stack: A/B test on the device + bench with per-kernel tables
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_lab_ab.py -q -x -m gpu 2>&1 | tail -5
timeout 300 python bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic > $O/r03_l.json 2> $O/r03_l.err
echo "stack: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_l.json) $(grep -o '"latency_b1[^}]*}' $O/r03_l.json | head -c 300)"
grep "stack\|total" $O/r03_l.err
MI355VITS_NO_DDS_STACK=1 timeout 300 python tools/lab_bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic > $O/r03_l2.json 2> $O/r03_l2.err
echo "lab pieces: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_l2.json)"; grep "total" $O/r03_l2.err
timeout 300 python tools/lab_bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic > $O/r03_l3.json 2> $O/r03_l3.err
echo "lab stack: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_l3.json)"; grep "total\|stack" $O/r03_l3.err
