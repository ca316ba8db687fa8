"""Why do bench.py's extra legs (fresh Workloads in the process that has already run the headline) land at ~9.6 ms whatever
their kernels cost?  Same process, same box: a mode's 2-handle step time (a) alone, (b) with the headline Workload still open,
(c) after it was closed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from mimic3_amd import weights as W
from mimic3_amd.config import VitsConfig

cfg = VitsConfig.apope_low()
weights = W.synthetic_weights(cfg, seed=1234)

def leg(math, streams=2, steps=60, warm=10):
    w = bench.Workload(cfg, weights, [0], streams, 32, 128, 6, 0, 1, math=math)
    w.size_workspaces()
    w.run_steps(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w.run_steps(steps)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps * 1e3
    return w, el

torch.cuda.set_device(0)
w, t = leg("f16x2"); print("f16x2 alone              %.3f ms" % t); w.close()
main, t = leg("bf16x3"); print("bf16x3 (kept open)       %.3f ms" % t)
w, t = leg("f16x2"); print("f16x2, bf16x3 still open %.3f ms" % t); w.close()
w, t = leg("bf16w"); print("bf16w, bf16x3 still open %.3f ms" % t); w.close()
main.close()
w, t = leg("f16x2"); print("f16x2, after close       %.3f ms" % t); w.close()
w, t = leg("bf16w"); print("bf16w, after close       %.3f ms" % t); w.close()
