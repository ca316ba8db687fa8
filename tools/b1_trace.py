#!/usr/bin/env python3
"""Batch-1 golden-shape calls in a loop (for `rocprofv3 --kernel-trace --stats -- python tools/b1_trace.py`): the sum of the pure kernel
durations per call against the call's wall time = what the 83 launches' dispatch gaps cost."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimic3_amd import weights as W  # noqa: E402
from mimic3_amd._native import Engine  # noqa: E402
from mimic3_amd.config import VitsConfig  # noqa: E402

cfg = VitsConfig.apope_low()
eng = Engine(W.pack(cfg, W.synthetic_weights(cfg, seed=1234)), device=0)
Txg = 180
ids = np.random.default_rng(99).integers(1, 50, (1, Txg)).astype(np.int64)
f1 = np.full((1, Txg), 5, np.int32)
f1[0, :91] = 6
N = int(os.environ.get("N", "200"))
for _ in range(20):
    eng.run(ids, [Txg], [0.667, 1.0, 0.8], forced_durations=f1, want_float=False, want_pcm16=True)
lat = []
for _ in range(N):
    t = time.perf_counter()
    eng.run(ids, [Txg], [0.667, 1.0, 0.8], forced_durations=f1, want_float=False, want_pcm16=True)
    lat.append(time.perf_counter() - t)
print("calls %d wall median %.4f ms min %.4f device %.4f" % (N + 20, 1e3 * float(np.median(lat)), 1e3 * min(lat), eng.last_run_ms()))
eng.close()
