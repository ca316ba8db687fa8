cd $GRAFT_REPO_ROOT
O=gpurun_out
for i in 1 2; do
timeout 300 python tools/lab_bench.py --steps 40 --streams 1 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r04_c.json 2> $O/r04_c.err
echo "run $i $(grep -o '"ms_per_step": [0-9.]*' $O/r04_c.json | head -1) $(grep "mrf_s\|mrf_p\|wn_layer" $O/r04_c.err | awk '{print $1, $4}' | tr '\n' ' ')"
done
MI355VITS_MRF_ABLATE=128 timeout 300 python tools/lab_bench.py --steps 12 --warmup 10 --streams 1 --no-extra --no-cpu-baseline --no-traffic --no-b1 --no-roofline > $O/r04_l.json 2> $O/r04_l.err
grep -a "mrf_s clocks" $O/r04_l.json | tail -6 | cut -c1-160
