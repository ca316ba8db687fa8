# round-3 evidence: GPU test suite, default bench line (all legs), rocprofv3 kernel stats + PMC passes of the bench workload
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > $O/r03_pytest_gpu_final.log 2>&1; grep "passed\|failed" $O/r03_pytest_gpu_final.log | tail -2
( time timeout 900 python bench.py > $O/r03_bench_final.json 2> $O/r03_bench_final.err ) 2>&1 | tail -3; head -2 $O/r03_bench_final.err
TAG=r03 bash tools/profile_r03.sh > $O/r03_profile.log 2>&1
python tools/rocprof_summary.py derived $O/r03_rocprof_pmc.txt > $O/r03_pmc_derived.txt 2>&1; head -12 $O/r03_pmc_derived.txt | cut -c1-200
for n in 1 2 3; do timeout 300 python bench.py --steps 60 --streams $n --no-extra --no-cpu-baseline --no-traffic --no-b1 > $O/r03_streams_$n.json 2> $O/r03_streams_$n.err; echo "streams $n: $(grep -o '"ms_per_step": [0-9.]*' $O/r03_streams_$n.json)"; done
STREAM=1 SECONDS=3 timeout 400 python tools/serve_bench.py > $O/r03_serve_bench.log 2> $O/r03_serve_bench.err; tail -8 $O/r03_serve_bench.log | cut -c1-330
