# a round's evidence run on one MI355X lease: GPU test suite, default bench line (all legs), rocprofv3 kernel stats + PMC passes of
# the bench workload, handles-in-flight sweep, serving shape.  TAG names the outputs (gpurun_out/${TAG}_*); copy what is to be
# judged into profiles/.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out
TAG=${TAG:-r06}
timeout 1700 python -m pytest tests -q -m gpu > $O/${TAG}_pytest_gpu_final.log 2>&1; grep "passed\|failed" $O/${TAG}_pytest_gpu_final.log | tail -2
( time timeout 900 python bench.py > $O/${TAG}_bench_final.json 2> $O/${TAG}_bench_final.err ) 2>&1 | tail -3; head -2 $O/${TAG}_bench_final.err
TAG=$TAG bash tools/profile.sh > $O/${TAG}_profile.log 2>&1
python tools/rocprof_summary.py derived $O/${TAG}_rocprof_pmc.txt > $O/${TAG}_pmc_derived.txt 2>&1; head -12 $O/${TAG}_pmc_derived.txt | cut -c1-200
for n in 1 2 3; do timeout 300 python bench.py --batch 32 --steps 60 --streams $n --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/${TAG}_streams_$n.json 2> $O/${TAG}_streams_$n.err; echo "streams $n: $(grep -o '"ms_per_step": [0-9.]*' $O/${TAG}_streams_$n.json | head -1)"; done
STREAM=1 SECONDS=3 timeout 400 python tools/serve_bench.py > $O/${TAG}_serve_bench.log 2> $O/${TAG}_serve_bench.err; tail -8 $O/${TAG}_serve_bench.log | cut -c1-330
# the driver's own invocation, for comparison with BENCH_rNN.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_args.json 2> $O/${TAG}_bench_driver_args.err; grep -o '"ms_per_step": [0-9.]*' $O/${TAG}_bench_driver_args.json | head -1
