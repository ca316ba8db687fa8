cd $GRAFT_REPO_ROOT
O=gpurun_out
for nw in 4 12 4 12; do
  MI355VITS_WN_WAVES=$nw timeout 300 python tools/lab_bench.py --steps 40 --streams 1 --no-extra --no-cpu-baseline --no-traffic > $O/r04_c.json 2> $O/r04_c.err
  echo "waves=$nw $(grep -o '"ms_per_step": [0-9.]*' $O/r04_c.json | head -1) $(grep "wn_layer" $O/r04_c.err | awk '{print $1, $4}' | tr '\n' ' ')"
done
