cd $GRAFT_REPO_ROOT
O=gpurun_out
for seg in "" 6240 3264 1632 12288; do
  if [ -n "$seg" ]; then export MI355VITS_MRF_SWEEP_SEG=$seg; else unset MI355VITS_MRF_SWEEP_SEG; fi
  timeout 300 python tools/lab_bench.py --steps 30 --streams 1 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r04_c.json 2> $O/r04_c.err
  echo "seg=[$seg] $(grep -o '"ms_per_step": [0-9.]*' $O/r04_c.json | head -1) $(grep "mrf_s" $O/r04_c.err | awk '{print $1, $4}' | tr '\n' ' ')"
done
