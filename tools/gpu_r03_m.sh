cd $GRAFT_REPO_ROOT
O=gpurun_out
for ab in 0 32 64 67 4 8 16 31; do
MI355VITS_DDS_ABLATE=$ab timeout 300 python tools/lab_bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_m.json 2> $O/r03_m.err
echo "ablate $ab: $(grep 'stack' $O/r03_m.err | tr '\n' ' ')"
done
