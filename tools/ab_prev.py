"""A/B of two builds of the PRODUCT library on one MI355X lease: bit identity first, then alternating bench runs.

    python tools/ab_prev.py bits  PREV.so NEW.so          # the same ragged batch through both: waveforms, lengths, PCM compared bitwise
    python tools/ab_prev.py bench PREV.so NEW.so [--batch 256 --steps 20 --rounds 2]

`bench` swaps the file bench.py opens (mimic3_amd/csrc/libmi355vits.so) — the product reads no library switch — and always puts NEW
back at the end.  Output: one line per run with ms/step and the kernel table rows named in --rows.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PRODUCT = os.path.join(ROOT, "mimic3_amd", "csrc", "libmi355vits.so")


def run_bits(lib_path, math=None):
    import numpy as np

    from mimic3_amd import weights as W
    from mimic3_amd._native import Engine, NativeLibrary
    from mimic3_amd.config import VitsConfig

    out = {}
    for voice in ("apope_low", "vctk_low"):
        cfg = getattr(VitsConfig, voice)()
        w = W.synthetic_weights(cfg, seed=11, frames_per_id=4.0)
        eng = Engine(W.pack(cfg, w), device=0, library=NativeLibrary(lib_path))
        if math is not None:
            eng.set_math(math)
        rng = np.random.default_rng(5)
        lens = [97, 128, 33, 5, 64, 120, 1, 77]
        ids = np.zeros((len(lens), 128), np.int64)
        for b, n in enumerate(lens):
            ids[b, :n] = rng.integers(1, cfg.num_symbols, size=n)
        sid = (np.arange(len(lens)) % max(cfg.n_speakers, 1)).astype(np.int64) if cfg.n_speakers > 1 else None
        r = eng.run(ids, lens, [0.667, 1.0, 0.8], sid=sid, seed=1234, want_pcm16=True)
        out[voice] = (r["audio"].copy(), r["lengths"].copy(), r["pcm"].copy())
        eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["bits", "bench"])
    ap.add_argument("prev")
    ap.add_argument("new")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--rows", default="flow.wn_layer_b3,dec.mrf_p.s2,dec.mrf_s.s1,dec.rb.s0,enc.ffn1,enc.ffn2,dec.conv_pre")
    a = ap.parse_args()
    if a.mode == "bits":
        import numpy as np

        ok = True
        for math in (None,):
            p, n = run_bits(a.prev, math), run_bits(a.new, math)
            for v in p:
                same = all(np.array_equal(x, y) for x, y in zip(p[v], n[v]))
                print(f"bits {v}: {'IDENTICAL' if same else 'DIFFERENT'} ({int(p[v][1].sum())} samples)")
                if not same:
                    d = np.abs(p[v][0].astype(np.float64) - n[v][0].astype(np.float64))
                    print("   max abs diff", d.max(), "lengths equal", np.array_equal(p[v][1], n[v][1]))
                ok = ok and same
        sys.exit(0 if ok else 1)
    keep = PRODUCT + ".ab_keep"
    shutil.copy2(a.new, keep)
    try:
        for r in range(a.rounds):
            for tag, lib in (("prev", a.prev), ("new", keep)):
                shutil.copy2(lib, PRODUCT)
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(a.batch), "--steps", str(a.steps), "--warmup", "5",
                       "--no-extra", "--no-cpu-baseline", "--no-traffic", "--no-b1"]
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                try:
                    d = json.loads(pr.stdout.strip().splitlines()[-1])
                    rf = d.get("roofline", {})
                    rows = "  ".join(f"{k}={rf.get('ms:' + k)}" for k in a.rows.split(","))
                    print(f"{tag} b{a.batch} round {r}: ms/step {d['ms_per_step']:.3f} | {rows}", flush=True)
                except Exception as ex:  # noqa: BLE001
                    print(f"{tag}: no bench line ({ex}); stderr tail: {pr.stderr[-400:]}", flush=True)
    finally:
        shutil.copy2(keep, PRODUCT)
        os.remove(keep)


if __name__ == "__main__":
    main()
