"""Randomised soak of the product library on one MI355X: ragged batches of random shapes, every batch checked row by row.

    python tools/gpu_soak.py [--seconds 240] [--seed 1] [--large-every 4]

Per iteration: a random voice (apope_low / vctk_low), batch size, phoneme counts, per-phoneme forced durations (so the rows' frame counts
are ragged too) and noise scales; the batch runs once, then
  * two random rows run ALONE (same seed, their own utterance index): lengths, float audio and int16 must be BITWISE the batched rows
    (the engine's batch semantics — DESIGN.md §1 — on shapes no fixed test has: the cursor decode of the ragged items, the WaveNet tile
    width chosen by grid, the micro-batched serving shapes all depend on B / T);
  * every 8th iteration one row is compared with the PyTorch-CPU oracle (rel. RMS <= 5e-6, equal length);
  * the int16 rows equal audio_float_to_int16 of the engine's own float rows.
Prints one line per failure and a summary; exit status 1 on any failure.  Test infrastructure (imports oracle/).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--large-every", type=int, default=0, help="every N-th batch is a large one (64 - 256 rows x 64 - 128 phonemes x up to 8 frames per phoneme)")
    a = ap.parse_args()
    from mimic3_amd import weights as W
    from mimic3_amd._native import Engine
    from mimic3_amd.config import VitsConfig
    from oracle.vits_oracle import VitsOracle, audio_float_to_int16

    rng = np.random.default_rng(a.seed)
    voices = {}
    for name in ("apope_low", "vctk_low"):
        cfg = getattr(VitsConfig, name)()
        w = W.synthetic_weights(cfg, seed=21, frames_per_id=3.0)
        voices[name] = (cfg, w, Engine(W.pack(cfg, w), device=0))
    t_end = time.time() + a.seconds
    it = fails = rows = orc = nlarge = 0
    frames = 0
    while time.time() < t_end:
        name = ("apope_low", "vctk_low")[it % 2]
        cfg, w, eng = voices[name]
        large = a.large_every > 0 and it % a.large_every == a.large_every - 1
        if large:  # grids on which the launchers pick their large-grid forms (128-column WaveNet tiles, the row-sweep MRF stage, the 128-column
            # encoder convs, the high-occupancy attention): the solo runs below take the small-grid forms of the same kernels — same bits
            B = int(rng.choice([64, 96, 128, 200, 256]))
            Tx = int(rng.choice([64, 100, 128]))
            nlarge += 1
        else:
            B = int(rng.choice([1, 2, 3, 5, 8, 13, 21, 32, 48]))
            Tx = int(rng.choice([1, 7, 33, 64, 65, 128, 129, 200]))
        lengths = rng.integers(1, Tx + 1, size=B)
        lengths[rng.integers(0, B)] = Tx
        ids = np.zeros((B, Tx), np.int64)
        for b in range(B):
            ids[b, : lengths[b]] = rng.integers(1, cfg.num_symbols, size=lengths[b])
        fmax = int(rng.choice([4, 6, 8])) if large else int(rng.choice([1, 2, 4, 8]))
        forced = rng.integers(1, fmax + 1, size=(B, Tx)).astype(np.int32)
        sid = rng.integers(0, cfg.n_speakers, size=B).astype(np.int64) if cfg.n_speakers > 1 else None
        scales = [float(rng.choice([0.0, 0.667])), 1.0, 0.8]
        seed = int(rng.integers(0, 1 << 30))
        full = eng.run(ids, lengths, scales, sid=sid, forced_durations=forced, seed=seed, want_pcm16=True)
        frames += int(full["lengths"].sum()) // 256
        for b in rng.choice(B, size=min(2, B), replace=False):
            b = int(b)
            one = eng.run(ids[b:b + 1], lengths[b:b + 1], scales, sid=None if sid is None else sid[b:b + 1], forced_durations=forced[b:b + 1],
                          seed=seed, utterance_base=b, want_pcm16=True)
            L = int(one["lengths"][0])
            ok = L == int(full["lengths"][b]) and np.array_equal(one["audio"][0, :L], full["audio"][b, :L]) and np.array_equal(one["pcm"][0, :L], full["pcm"][b, :L])
            ok = ok and np.array_equal(full["pcm"][b, :L], audio_float_to_int16(full["audio"][b, :L]))
            rows += 1
            if not ok:
                fails += 1
                print(f"FAIL batched != alone: iter {it} {name} B={B} Tx={Tx} row {b} len {lengths[b]} fmax {fmax} L {L} vs {int(full['lengths'][b])}", flush=True)
        if it % 8 == 0 and scales[0] == 0.0:
            b = int(rng.integers(0, B))
            # (deterministic scales on both sides: the oracle has no Philox stream)
            one = eng.run(ids[b:b + 1], lengths[b:b + 1], [0.0, 1.0, 0.0], sid=None if sid is None else sid[b:b + 1], forced_durations=forced[b:b + 1], seed=seed)
            ref = VitsOracle(cfg, w).infer(ids[b:b + 1], lengths[b:b + 1], [0.0, 1.0, 0.0], sid=None if sid is None else sid[b:b + 1], forced_durations=forced[b:b + 1])
            L = int(one["lengths"][0])
            Lr = int(ref["audio_lengths"][0])
            x, r = one["audio"][0, :L].astype(np.float64), ref["audio"][0, 0, :Lr].astype(np.float64)
            rel = float(np.sqrt(np.mean((x - r) ** 2)) / max(np.sqrt(np.mean(r ** 2)), 1e-30)) if L == Lr else float("inf")
            orc += 1
            if not rel <= 5e-6:
                fails += 1
                print(f"FAIL vs oracle: iter {it} {name} Tx={Tx} len {lengths[b]} L {L} / {Lr} rel {rel:.3e}", flush=True)
        it += 1
    for _, _, e in voices.values():
        e.close()
    print(f"soak: {it} batches ({nlarge} large), {rows} rows checked batched == alone (bitwise), {orc} rows vs the oracle, {frames} frames synthesised, {fails} failures", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
