# WaveNet layer phase ablations on the lab build (MI355VITS_WN_ABLATE: 1 no MFMA loops, 2 no staging, 4 no epilogue)
cd $GRAFT_REPO_ROOT
O=gpurun_out
for ab in 0 1 2 4 3 5 6 7; do
  MI355VITS_WN_ABLATE=$ab timeout 300 python tools/lab_bench.py --steps 30 --streams 1 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_k.json 2> $O/r03_k.err
  echo "wn ablate $ab: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_k.json)"; grep "flow.wn_layer_b3" $O/r03_k.err | tail -1
done
