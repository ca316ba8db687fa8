cd $GRAFT_REPO_ROOT
O=gpurun_out
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 40 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_j.json 2> $O/r03_j.err
  echo "rep $rep: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_j.json)"; grep "dec.mrf_p" $O/r03_j.err
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -2
