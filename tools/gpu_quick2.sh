cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-extra > gpurun_out/q2_bench.json 2> gpurun_out/q2_bench.err; grep -B2 -A12 "^per-kernel" gpurun_out/q2_bench.err | head -40; head -3 gpurun_out/q2_bench.err
timeout 200 python bench.py --steps 30 --streams 1 --no-cpu-baseline --no-extra --no-b1 2>&1 >/dev/null | grep "headline\|wn_layer\|upsample"
