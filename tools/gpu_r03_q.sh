cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_lab_ab.py tests/test_gpu_serving.py -q -x -m gpu -k "flow_pointwise or slice" 2>&1 | tail -3
for rep in 1 2; do
timeout 300 python bench.py --steps 60 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_q.json 2> $O/r03_q.err
echo "slice: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_q.json)"; grep "flow\.p" $O/r03_q.err
MI355VITS_NO_FLOW_GEMM=1 timeout 300 python tools/lab_bench.py --steps 60 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_q2.json 2> $O/r03_q2.err
echo "lab general: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_q2.json)"; grep "flow\.p" $O/r03_q2.err
timeout 300 python tools/lab_bench.py --steps 60 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_q3.json 2> $O/r03_q3.err
echo "lab slice: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_q3.json)"
done
