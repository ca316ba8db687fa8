#!/bin/bash
# Evidence run on the GPU box (everything lands in gpurun_out/):
#   1. the default bench line (with cpu_baseline)                    -> bench_default.json / .err
#   2. rocprofv3 --kernel-trace --stats of the bench (one stream)    -> prof_stats/  + rocprof_stats.txt
#   3. rocprofv3 --pmc passes, counters only (separate runs)         -> prof_pmc{1,2,3}/ + rocprof_pmc.txt
# rocprofv3 wants a writable cwd and TMPDIR.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.json
BENCH="python $R/bench.py --steps 4 --warmup 2 --streams 1 --no-cpu-baseline --no-roofline --no-b1"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_pmc1 $R/gpurun_out/prof_pmc2 $R/gpurun_out/prof_pmc3
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o stats -- $BENCH > $R/gpurun_out/rocprof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_pmc1 -o pmc1 -- $BENCH > $R/gpurun_out/rocprof_pmc1.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_pmc2 -o pmc2 -- $BENCH > $R/gpurun_out/rocprof_pmc2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
    -d $R/gpurun_out/prof_pmc3 -o pmc3 -- $BENCH > $R/gpurun_out/rocprof_pmc3.log 2>&1
cd $R
( echo "# rocprofv3 --kernel-trace --stats -- $BENCH"; python tools/rocprof_summary.py stats gpurun_out/prof_stats ) > gpurun_out/rocprof_stats.txt 2>&1
( echo "# rocprofv3 --pmc <one counter set per run> -- $BENCH   (per-launch means, counters summed over instances)"; python tools/rocprof_summary.py pmc gpurun_out/prof_pmc1 gpurun_out/prof_pmc2 gpurun_out/prof_pmc3 ) > gpurun_out/rocprof_pmc.txt 2>&1
python tools/rocprof_summary.py traffic gpurun_out/pmc_traffic.json gpurun_out/prof_pmc1 gpurun_out/prof_pmc2 > /dev/null 2>&1
# keep the merge small: the summaries are what gets committed
find gpurun_out/prof_stats gpurun_out/prof_pmc1 gpurun_out/prof_pmc2 gpurun_out/prof_pmc3 -name "*.db" -size +30M -delete
head -12 gpurun_out/rocprof_stats.txt | cut -c1-150
