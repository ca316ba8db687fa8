#!/bin/bash
# one GPU box visit: parity tests, then the bench line with the per-kernel table (everything into gpurun_out/)
mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench.log 2>&1
grep -E "ms/step" gpurun_out/bench.log | head -${GPU_QUICK_ROWS:-12}; tail -1 gpurun_out/bench.log | cut -c1-330
