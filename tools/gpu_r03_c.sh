# lab build (scalar VALU in k_mrf_p) vs product (packed), and phase ablations of k_mrf_p: 1 = no MFMA loops, 2 = no staging, 4 = no stores
cd $GRAFT_REPO_ROOT
O=gpurun_out
for v in 0 1 2 4 7; do
  MI355VITS_MRF_ABLATE=$v timeout 300 python tools/lab_bench.py --steps 20 --no-extra --no-cpu-baseline --no-b1 > $O/r03_labc_$v.json 2> $O/r03_labc_$v.err
  echo "lab (scalar) ablate $v"; grep "dec.mrf_p" $O/r03_labc_$v.err
done
timeout 300 python bench.py --steps 20 --no-extra --no-cpu-baseline --no-b1 > $O/r03_bench_c.json 2> $O/r03_bench_c.err; echo product; grep "dec.mrf" $O/r03_bench_c.err
