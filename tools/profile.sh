# rocprofv3 kernel-trace + PMC passes of the bench workload (one stream, no extra legs); TAG names the outputs
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
TAG=${TAG:-run}
BENCH="python $R/bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-extra --no-b1 --no-roofline --no-traffic ${MATH:+--math $MATH} ${BATCH:+--batch $BATCH}"
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq1 $O/prof_sq2 $O/prof_sq3 $O/prof_sq4
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats -- $BENCH > $O/rocprof_stats.log 2>&1
if [ -z "$STATS_ONLY" ]; then
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -- $BENCH > $O/rocprof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -- $BENCH > $O/rocprof_write.log 2>&1
fi
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/prof_sq1 -- $BENCH > $O/rocprof_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/prof_sq2 -- $BENCH > $O/rocprof_sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d $O/prof_sq3 -- $BENCH > $O/rocprof_sq3.log 2>&1
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH GRBM_GUI_ACTIVE -d $O/prof_sq4 -- $BENCH > $O/rocprof_sq4.log 2>&1
cd $R
python tools/rocprof_summary.py stats gpurun_out/prof_stats > gpurun_out/${TAG}_rocprof_stats.txt 2>&1
DIRS="gpurun_out/prof_sq1 gpurun_out/prof_sq2 gpurun_out/prof_sq3 gpurun_out/prof_sq4"
if [ -z "$STATS_ONLY" ]; then DIRS="gpurun_out/prof_fetch gpurun_out/prof_write $DIRS"; fi
python tools/rocprof_summary.py pmc $DIRS > gpurun_out/${TAG}_rocprof_pmc.txt 2>&1
if [ -z "$STATS_ONLY" ]; then python tools/rocprof_summary.py traffic gpurun_out/${TAG}_pmc_traffic.json gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/${TAG}_traffic.log 2>&1; fi
head -16 gpurun_out/${TAG}_rocprof_stats.txt | cut -c1-160; head -12 gpurun_out/${TAG}_rocprof_pmc.txt | cut -c1-700
tail -3 $O/rocprof_sq3.log
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq1 gpurun_out/prof_sq2 gpurun_out/prof_sq3 gpurun_out/prof_sq4
