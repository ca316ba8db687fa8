#!/usr/bin/env python3
"""Waveform error of each math mode against the fp64 oracle (same weights, same inputs, both Gaussian draws injected):
what the operand representation costs end to end.  Run on the GPU box; prints one JSON line per mode."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mimic3_amd import weights as W  # noqa: E402
from mimic3_amd._native import Engine  # noqa: E402
from mimic3_amd.config import VitsConfig  # noqa: E402
from oracle.vits_oracle import VitsOracle  # noqa: E402  (checker)


def main():
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234, frames_per_id=6.0)
    B, Tx, fpi = 2, 64, 6
    rng = np.random.default_rng(4321)
    ids = rng.integers(1, 50, (B, Tx)).astype(np.int64)
    lengths = np.array([Tx, Tx - 9], np.int64)
    forced = np.full((B, Tx), fpi, np.int32)
    nw = rng.standard_normal((B, 2, Tx)).astype(np.float32)
    nz = rng.standard_normal((B, cfg.inter_channels, Tx * fpi)).astype(np.float32)
    scales = (0.667, 1.0, 0.8)
    ref = {}
    for name, dt in (("fp64", torch.float64), ("fp32", torch.float32)):
        r = VitsOracle(cfg, w, dtype=dt).infer(ids, lengths, scales, noise_w=nw, noise_z=nz, forced_durations=forced, stage_rows=())
        ref[name] = [np.asarray(r["audio"][b, 0, : int(r["audio_lengths"][b])], np.float64) for b in range(B)]
    rel = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    print(json.dumps({"mode": "oracle fp32 (PyTorch CPU)", "rel_rms_vs_fp64": max(rel(ref["fp32"][b], ref["fp64"][b]) for b in range(B))}))
    eng = Engine(W.pack(cfg, w))
    for mode in ("f32", "bf16x3", "f16x2", "bf16w"):
        eng.set_math(mode)
        out = eng.run(ids, lengths, scales, noise_w=nw, noise_z=nz, forced_durations=forced)
        errs = [rel(out["audio"][b, : int(out["lengths"][b])].astype(np.float64), ref["fp64"][b]) for b in range(B)]
        print(json.dumps({"mode": mode, "rel_rms_vs_fp64": max(errs), "rows": errs}))
    eng.close()


if __name__ == "__main__":
    main()
