cd $GRAFT_REPO_ROOT
O=gpurun_out
export MI355VITS_MRF_SWEEP_SEG=6240
for ab in 0 2 4 6 1; do
  MI355VITS_MRF_ABLATE=$ab timeout 300 python tools/lab_bench.py --steps 30 --streams 1 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r04_c.json 2> $O/r04_c.err
  echo "ablate=$ab $(grep "mrf_s" $O/r04_c.err | awk '{print $1, $4}' | tr '\n' ' ')"
done
