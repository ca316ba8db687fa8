cd $GRAFT_REPO_ROOT
O=gpurun_out
for v in 0 8 16 32 56 57 63; do
  MI355VITS_MRF_ABLATE=$v timeout 300 python tools/lab_bench.py --steps 20 --no-extra --no-cpu-baseline --no-b1 > $O/r03_labd_$v.json 2> $O/r03_labd_$v.err
  echo "lab ablate $v"; grep "dec.mrf_p" $O/r03_labd_$v.err
done
