# A/B of the symmetry breakers in k_mrf_p (lab build): MI355VITS_MRF_ABLATE bits 0x100 / 0x200 / 0x400
cd $GRAFT_REPO_ROOT
O=gpurun_out
for v in 0 0x100 0x200 0x400; do
  MI355VITS_MRF_ABLATE=$v timeout 300 python tools/lab_bench.py --steps 30 --no-extra --no-cpu-baseline --no-b1 > $O/r03_lab_$v.json 2> $O/r03_lab_$v.err
  echo "variant $v"; grep "ms_per_step" $O/r03_lab_$v.json | sed 's/.*"ms_per_step": \([0-9.]*\).*/step \1 ms/'; grep "dec.mrf" $O/r03_lab_$v.err
done
timeout 300 python bench.py --steps 30 --no-extra --no-cpu-baseline --no-b1 > $O/r03_bench_b.json 2> $O/r03_bench_b.err; echo product; grep "dec.mrf" $O/r03_bench_b.err
