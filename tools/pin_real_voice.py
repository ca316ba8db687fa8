#!/usr/bin/env python3
"""Real-voice pin harness: the moment a Mimic 3 voice (``generator.onnx`` + ``config.json``) is available, this settles
the two details the oracle could not pin offline (ResBlock2 and the export's ``scales`` plumbing).

    python tools/pin_real_voice.py VOICE_DIR --ids ids.json [--wav apope_sample_amd64.wav] [--key en_UK/apope_low]

What it does, in the order the reference's own acceptance test does it (``tests/apope_sample.txt`` ->
``mimic3 --deterministic`` -> ``tests/samples_match.py`` against ``tests/apope_sample_<arch>.wav``, Dockerfile:99-105):

1. sha256 + size of ``generator.onnx`` against the published catalogue (``mimic3_tts/voices.json``; the entries of the
   two benchmark voices are restated in ``KNOWN_VOICES`` below);
2. loads the file through ``mimic3_amd.onnx_import`` (the product's importer) and builds the engine on it;
3. feeds the phoneme ids of the sample sentence (``--ids``: a JSON list — text -> ids needs gruut, which lives above the
   boundary; any machine with the reference installed prints them with ``mimic3 --voice … --deterministic`` debug logging,
   or via ``Mimic3Voice.text_to_ids``) with ``scales = [0, length_scale, 0]`` — ``--deterministic`` zeroes both noise
   scales (``mimic3_tts/__main__.py:224-228``);
4. applies ``samples_match.py``'s criterion (at most 10 % of the int16 samples differ, a length difference counting as
   differing samples; ``tests/samples_match.py:35-59``) between the ENGINE's int16 and the golden WAV, and between the
   ORACLE's int16 (same imported weights) and the golden WAV: ResBlock2 and the feed plumbing are then pinned on both
   sides;
5. compares engine and oracle directly (rel. RMS of the float waveform, durations exactly equal).

The oracle is the checker here (test infrastructure); nothing in ``mimic3_amd`` imports this file.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# mimic3_tts/voices.json ("files" -> size_bytes / sha256_sum) for the voices BASELINE.json names
KNOWN_VOICES = {
    "en_UK/apope_low": {
        "generator.onnx": (62792219, "0b5a323500ebd022351db12da2b3aab8cdd47d0826d173e780a58b93604618c9"),
        "config.json": (3434, "1fdaa1124e02cc177eb776fbc6e08c838b56bd2e86c82d8d7fe434d9337806b0"),
        "phonemes.txt": (263, "8f9c3e6ced14d7fc5426e4e1bc7f7cc1037a20a645ca34110abcb76148fa8bfd"),
    },
    "en_US/vctk_low": {
        "generator.onnx": (76546145, "c958303de83a59fac937a91009c9081b5f2f7369890b9969e05141e56e867d2b"),
        "config.json": (3555, "ab38b8df74db751dc89d43c17f238ee7a5e56d8e26f59673e272ea4802d275a7"),
        "phonemes.txt": (263, "8f9c3e6ced14d7fc5426e4e1bc7f7cc1037a20a645ca34110abcb76148fa8bfd"),
    },
}
GOLDEN_SAMPLES = 253696  # tests/apope_sample_{amd64,arm64,armv7}.wav: 991 frames x 256


def sha256_file(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


def check_catalogue(voice_dir: str, key: str) -> dict:
    """File sizes + sha256 against the published catalogue; raises on mismatch."""
    rep = {}
    for name, (size, digest) in KNOWN_VOICES[key].items():
        p = os.path.join(voice_dir, name)
        if not os.path.isfile(p):
            if name == "generator.onnx":
                raise FileNotFoundError(p)
            continue
        got = (os.path.getsize(p), sha256_file(p))
        rep[name] = {"size": got[0], "sha256": got[1], "matches_catalogue": got == (size, digest)}
        if got != (size, digest):
            raise ValueError(f"{p}: size/sha256 {got} differ from voices.json {(size, digest)} for {key}")
    return rep


def samples_match_fraction(a: np.ndarray, b: np.ndarray) -> float:
    """The reference's criterion (tests/samples_match.py:35-59): number of differing int16 samples over the shorter
    length, a length difference counting as that many differing samples."""
    a = np.asarray(a, np.int16).reshape(-1)
    b = np.asarray(b, np.int16).reshape(-1)
    n = min(a.size, b.size)
    if n == 0:
        raise ValueError("Empty WAV")
    return (abs(a.size - b.size) + int(np.count_nonzero(a[:n] != b[:n]))) / n


def read_wav_int16(path: str) -> np.ndarray:
    with wave.open(path, "rb") as w:
        if (w.getsampwidth(), w.getnchannels()) != (2, 1):
            raise ValueError(f"{path}: expected 16-bit mono")
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.int16)


def pin(voice_dir: str, ids, wav_path=None, key=None, library=None, sid=None, percent_threshold: float = 0.10) -> dict:
    """Run steps 1-5; returns the report (raises AssertionError / ValueError on a failed pin)."""
    from mimic3_amd import onnx_import
    from mimic3_amd import weights as W
    from mimic3_amd._native import Engine
    from oracle.vits_oracle import VitsOracle, audio_float_to_int16

    onnx_path = os.path.join(voice_dir, "generator.onnx")
    report = {"voice_dir": voice_dir, "key": key}
    if key is not None:
        report["catalogue"] = check_catalogue(voice_dir, key)
    else:
        report["generator_sha256"] = sha256_file(onnx_path)
    cfg, tensors = onnx_import.import_onnx(onnx_path)
    report["config"] = {"resblock": cfg.resblock, "upsample_rates": list(cfg.upsample_rates), "n_speakers": cfg.n_speakers,
                        "num_symbols": cfg.num_symbols, "parameters": int(sum(int(np.prod(t.shape)) for t in tensors.values()))}
    length_scale = 1.0
    cj = os.path.join(voice_dir, "config.json")
    if os.path.isfile(cj):
        with open(cj) as f:
            length_scale = float(json.load(f).get("inference", {}).get("length_scale", 1.0))
    ids = np.asarray(ids, dtype=np.int64).reshape(1, -1)
    lengths = np.array([ids.shape[1]], np.int64)
    scales = np.array([0.0, length_scale, 0.0], np.float32)  # --deterministic: both noise scales 0
    sid_a = None
    if cfg.is_multispeaker:
        sid_a = np.array([0 if sid is None else int(sid)], np.int64)
    eng = Engine(W.pack(cfg, tensors), library=library)
    out = eng.run(ids, lengths, scales, sid_a, want_float=True, want_pcm16=True)
    eng.close()
    ora = VitsOracle(cfg, tensors).infer(ids, lengths, scales, sid=sid_a)
    L = int(out["lengths"][0])
    assert L == int(ora["audio_lengths"][0]), ("durations differ", L, int(ora["audio_lengths"][0]))
    a = out["audio"][0, :L].astype(np.float64)
    r = ora["audio"][0, 0, :L].astype(np.float64)
    rel = float(np.sqrt(np.mean((a - r) ** 2)) / max(1e-30, np.sqrt(np.mean(r ** 2))))
    eng16 = out["pcm"][0, :L]
    ora16 = audio_float_to_int16(ora["audio"][0, 0, :L])
    report["engine_vs_oracle"] = {"samples": L, "rel_rms": rel, "int16_fraction_differing": samples_match_fraction(eng16, ora16)}
    assert rel < 1e-4, ("engine vs oracle on the real weights", rel)
    if wav_path:
        gold = read_wav_int16(wav_path)
        fe, fo = samples_match_fraction(eng16, gold), samples_match_fraction(ora16, gold)
        report["golden_wav"] = {"path": wav_path, "samples": int(gold.size), "engine_fraction_differing": fe,
                                "oracle_fraction_differing": fo, "threshold": percent_threshold,
                                "engine_max_lsb": int(np.abs(eng16[: gold.size].astype(np.int32) - gold[:L].astype(np.int32)).max())
                                if min(L, gold.size) else None}
        assert fe <= percent_threshold, ("engine vs golden WAV (samples_match.py criterion)", fe)
        assert fo <= percent_threshold, ("oracle vs golden WAV: the restatement of ResBlock2 / the feed is wrong", fo)
        report["pinned"] = True
    else:
        report["pinned"] = False  # nothing external to pin against: engine == oracle on real weights only
    return report


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("voice_dir")
    ap.add_argument("--ids", required=True, help="JSON list of phoneme ids of the sample sentence")
    ap.add_argument("--wav", help="golden WAV (tests/apope_sample_amd64.wav)")
    ap.add_argument("--key", choices=sorted(KNOWN_VOICES), help="check sizes / sha256 against voices.json")
    ap.add_argument("--sid", type=int)
    ap.add_argument("--emu", action="store_true", help="CPU model of the kernels instead of the HIP library (no GPU here)")
    a = ap.parse_args(argv)
    with open(a.ids) as f:
        ids = json.load(f)
    lib = None
    if a.emu:
        from mimic3_amd import build
        from mimic3_amd._native import NativeLibrary

        lib = NativeLibrary(build.build_emu())
    rep = pin(a.voice_dir, ids, a.wav, a.key, lib, a.sid)
    print(json.dumps(rep, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
