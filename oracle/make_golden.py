#!/usr/bin/env python3
"""Freeze oracle outputs (and the reference's own int16 conversion) into tests/golden/.

TEST INFRASTRUCTURE.  Run in the build container from the repo root: ``python oracle/make_golden.py``.
* ``int16_reference.npz``: inputs/outputs of the *reference's* ``audio_float_to_int16``
  (``/root/reference/mimic3_tts/utils.py:237-244`` imported by path) — pins A2 of the oracle and engine.
* ``oracle_apope_low_b1.npz``: oracle waveform for the en_UK/apope_low graph on seeded ids (weights are
  re-derived from the seed; a checksum guards against RNG drift).
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimic3_amd import weights as W  # noqa: E402
from mimic3_amd.config import VitsConfig  # noqa: E402
from oracle.vits_oracle import VitsOracle, audio_float_to_int16  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(G, exist_ok=True)
    ref_utils = "/root/reference/mimic3_tts/utils.py"
    if os.path.exists(ref_utils):
        spec = importlib.util.spec_from_file_location("m3utils", ref_utils)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        rng = np.random.default_rng(0)
        ins, outs = [], []
        cases = [np.array([0, .5, -.25, 1e-4], np.float32), np.zeros(7, np.float32), np.array([1e-5, -2e-5], np.float32),
                 np.array([1.0, -1.0, 0.999999], np.float32)]
        for i in range(40):
            n = int(rng.integers(1, 4000))
            cases.append(np.tanh(rng.standard_normal(n) * rng.choice([1e-4, 1e-2, 0.3, 1.0, 3.0])).astype(np.float32))
        for a in cases:
            r = m.audio_float_to_int16(a)
            assert np.array_equal(r, audio_float_to_int16(a))
            ins.append(a)
            outs.append(r)
        np.savez_compressed(os.path.join(G, "int16_reference.npz"), n=np.array(len(ins)),
                            **{f"in{i}": a for i, a in enumerate(ins)}, **{f"out{i}": a for i, a in enumerate(outs)})
        print("int16_reference.npz:", len(ins), "cases from the reference's audio_float_to_int16")
    cfg = VitsConfig.apope_low()
    seed, fpi = 4321, 3.0
    w = W.synthetic_weights(cfg, seed=seed, frames_per_id=fpi)
    ids = np.random.default_rng(5).integers(1, cfg.num_symbols, size=(1, 28)).astype(np.int64)
    lengths = np.array([28], np.int64)
    r = VitsOracle(cfg, w).infer(ids, lengths, [0, 1, 0])
    audio = r["audio"][:, 0]
    np.savez_compressed(os.path.join(G, "oracle_apope_low_b1.npz"), config_json=np.array(cfg.to_json()), seed=np.array(seed),
                        frames_per_id=np.array(fpi), ids=ids, lengths=lengths, audio=audio.astype(np.float32),
                        pcm=np.stack([audio_float_to_int16(audio[0])]), audio_lengths=r["audio_lengths"],
                        w_ceil=r["w_ceil"].astype(np.int32),
                        weight_checksum=np.array(float(sum(float(np.abs(v).sum()) for v in w.values()))))
    print("oracle_apope_low_b1.npz:", audio.shape, r["audio_lengths"])


if __name__ == "__main__":
    main()
