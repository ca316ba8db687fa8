set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
# MATH=f16x2 bash tools/profile_r02.sh profiles another math mode (default: the engine's)
BENCH="python $R/bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-extra --no-b1 --no-roofline ${MATH:+--math $MATH}"
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq1 $O/prof_sq2
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats -- $BENCH > $O/rocprof_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -- $BENCH > $O/rocprof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -- $BENCH > $O/rocprof_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/prof_sq1 -- $BENCH > $O/rocprof_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/prof_sq2 -- $BENCH > $O/rocprof_sq2.log 2>&1
cd $R
python tools/rocprof_summary.py stats gpurun_out/prof_stats > gpurun_out/r02_rocprof_stats.txt 2>&1
python tools/rocprof_summary.py pmc gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq1 gpurun_out/prof_sq2 > gpurun_out/r02_rocprof_pmc.txt 2>&1
python tools/rocprof_summary.py traffic gpurun_out/r02_pmc_traffic.json gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/r02_traffic.log 2>&1
head -30 gpurun_out/r02_rocprof_stats.txt; head -20 gpurun_out/r02_rocprof_pmc.txt; cat gpurun_out/r02_traffic.log | head -40
du -sh gpurun_out/prof_* | head; rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq1 gpurun_out/prof_sq2
