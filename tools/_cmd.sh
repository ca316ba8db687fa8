cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python tools/lab_bench.py --steps 40 --streams 1 --no-extra --no-cpu-baseline --no-traffic > $O/r04_c.json 2> $O/r04_c.err
echo "$(grep -o '"ms_per_step": [0-9.]*' $O/r04_c.json | head -1) $(grep "wn_layer" $O/r04_c.err | awk '{print $1, $4}' | tr '\n' ' ')"
MI355VITS_WN_ABLATE=64 timeout 300 python tools/lab_bench.py --steps 30 --warmup 20 --streams 1 --no-extra --no-cpu-baseline --no-traffic --no-b1 --no-roofline > $O/r04_h.json 2> $O/r04_h.err
grep -a "wn phases" $O/r04_h.json | tail -4 | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bench_workload or golden" > $O/r04_i_pytest.log 2>&1; tail -2 $O/r04_i_pytest.log
