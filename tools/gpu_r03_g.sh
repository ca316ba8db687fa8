cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python bench.py --steps 30 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_bench_q.json 2> $O/r03_bench_q.err; echo "product (k_mrf_q)"; grep "ms_per_step" $O/r03_bench_q.json | sed 's/.*"ms_per_step": \([0-9.]*\).*/step \1 ms/'; grep "dec.mrf" $O/r03_bench_q.err
MI355VITS_NO_MRF_Q=1 timeout 300 python tools/lab_bench.py --steps 30 --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_bench_noq.json 2> $O/r03_bench_noq.err; echo "lab, k_mrf_p"; grep "dec.mrf" $O/r03_bench_noq.err
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "bench or golden or batched or mrf or vctk" 2>&1 | tail -3
