cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python bench.py --steps 30 --no-extra --no-cpu-baseline > $O/r03_bench_d.json 2> $O/r03_bench_d.err; echo product; grep "ms_per_step" $O/r03_bench_d.json | sed 's/.*"ms_per_step": \([0-9.]*\).*/step \1 ms/'; grep "dec.mrf\|headline\|total" $O/r03_bench_d.err
for v in 0 8 16 32 57 63; do
  MI355VITS_MRF_ABLATE=$v timeout 300 python tools/lab_bench.py --steps 20 --no-extra --no-cpu-baseline --no-b1 > $O/r03_labd_$v.json 2> $O/r03_labd_$v.err
  echo "lab ablate $v"; grep "dec.mrf_p" $O/r03_labd_$v.err
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -3
