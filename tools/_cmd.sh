cd $GRAFT_REPO_ROOT
O=gpurun_out
for seg in 0 24576 12288 8192 16384 6144 0 12288; do
  MI355VITS_MRF_SWEEP_SEG32=$seg timeout 300 python tools/lab_bench.py --steps 30 --streams 1 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r04_c.json 2> $O/r04_c.err
  echo "seg32=$seg $(grep "mrf_s\|mrf_p\|wn_layer" $O/r04_c.err | awk '{print $1, $4}' | tr '\n' ' ')"
done
