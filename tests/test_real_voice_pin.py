"""The real-voice pin harness (tools/pin_real_voice.py).

* Always (CPU): the harness end to end on a stand-in voice — a ``generator.onnx`` written by ``torch.onnx.export``
  with folded weight-norm (what the Mimic 3 trainer ships), ``config.json`` beside it, a "golden WAV" produced by the
  oracle — through the CPU model of the kernels; plus the failure modes (wrong hash, wrong audio).
* With ``MI355VITS_VOICE_DIR`` (a downloaded ``en_UK/apope_low`` directory) — and ``MI355VITS_SAMPLE_IDS`` (JSON list of
  the ids of tests/apope_sample.txt), ``MI355VITS_SAMPLE_WAV`` (tests/apope_sample_amd64.wav): the real pin, on the HIP
  library when a GPU is present.  Skipped while no voice file is reachable (no network in the build environment).
"""
import json
import os
import sys
import wave

import numpy as np
import pytest

from mimic3_amd import weights as W
from mimic3_amd.config import VitsConfig
from oracle.vits_oracle import VitsOracle, audio_float_to_int16
from tests.onnx_fixture import export_onnx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pin_real_voice as PIN  # noqa: E402


def _write_wav(path, pcm):
    with wave.open(str(path), "wb") as w:
        w.setframerate(22050)
        w.setsampwidth(2)
        w.setnchannels(1)
        w.writeframes(np.asarray(pcm, "<i2").tobytes())


def test_harness_end_to_end_on_an_exported_stand_in_voice(emu_lib, tmp_path):
    cfg = VitsConfig.tiny()  # ResBlock2 decoder, like the _low voices
    w = W.synthetic_weights(cfg, seed=77, frames_per_id=3.0)
    d = tmp_path / "en_UK" / "standin_low"
    d.mkdir(parents=True)
    (d / "generator.onnx").write_bytes(export_onnx(cfg, w, weight_norm_prefixes=("flow.", "dec.")))
    conf = json.loads(cfg.to_json())
    conf["inference"] = {"length_scale": 1.1, "noise_scale": 0.667, "noise_w": 0.8}
    (d / "config.json").write_text(json.dumps(conf))
    ids = [1, 5, 9, 3, 1, 7, 7, 2, 1]
    ora = VitsOracle(cfg, w).infer(np.array([ids]), np.array([len(ids)]), [0.0, 1.1, 0.0])
    gold = audio_float_to_int16(ora["audio"][0, 0, : int(ora["audio_lengths"][0])])
    _write_wav(tmp_path / "gold.wav", gold)
    (tmp_path / "ids.json").write_text(json.dumps(ids))

    rep = PIN.pin(str(d), ids, str(tmp_path / "gold.wav"), library=emu_lib)
    assert rep["pinned"] and rep["config"]["resblock"] == "2"
    assert rep["engine_vs_oracle"]["rel_rms"] < 1e-4 and rep["golden_wav"]["engine_fraction_differing"] <= 0.10
    assert rep["golden_wav"]["samples"] == gold.size

    # without a WAV the harness still compares engine and oracle on the imported weights, but says "not pinned"
    assert PIN.pin(str(d), ids, None, library=emu_lib)["pinned"] is False
    # a golden WAV of something else fails the reference's criterion
    _write_wav(tmp_path / "other.wav", np.roll(gold, 7) // 2)
    with pytest.raises(AssertionError, match="golden WAV"):
        PIN.pin(str(d), ids, str(tmp_path / "other.wav"), library=emu_lib)
    # the catalogue check refuses a file that is not the published one
    with pytest.raises(ValueError, match="voices.json"):
        PIN.pin(str(d), ids, None, key="en_UK/apope_low", library=emu_lib)
    # command line (what INTEGRATION.md documents)
    assert PIN.main([str(d), "--ids", str(tmp_path / "ids.json"), "--wav", str(tmp_path / "gold.wav"), "--emu"]) == 0


def test_samples_match_criterion_restated():
    """tests/samples_match.py:35-59: differing samples / shorter length, a length difference counts as differing."""
    a = np.arange(100, dtype=np.int16)
    b = a.copy()
    b[:7] += 1
    assert PIN.samples_match_fraction(a, b) == 0.07
    assert PIN.samples_match_fraction(a, a[:90]) == pytest.approx(10 / 90)
    with pytest.raises(ValueError):
        PIN.samples_match_fraction(a, a[:0])


def _real_pin(library):
    vd = os.environ["MI355VITS_VOICE_DIR"]
    with open(os.environ["MI355VITS_SAMPLE_IDS"]) as f:
        ids = json.load(f)
    rep = PIN.pin(vd, ids, os.environ.get("MI355VITS_SAMPLE_WAV"), key=os.environ.get("MI355VITS_VOICE_KEY", "en_UK/apope_low"),
                  library=library)
    print(json.dumps(rep, indent=1))
    if os.environ.get("MI355VITS_SAMPLE_WAV"):
        assert rep["pinned"] and rep["golden_wav"]["samples"] == PIN.GOLDEN_SAMPLES


needs_voice = pytest.mark.skipif(not (os.environ.get("MI355VITS_VOICE_DIR") and os.environ.get("MI355VITS_SAMPLE_IDS")),
                                 reason="no real voice reachable: set MI355VITS_VOICE_DIR (+ MI355VITS_SAMPLE_IDS, MI355VITS_SAMPLE_WAV)")


@needs_voice
def test_real_voice_pin_on_the_cpu_model(emu_lib):
    _real_pin(emu_lib)


@needs_voice
@pytest.mark.gpu
def test_real_voice_pin_on_the_hip_library(gpu_lib):
    _real_pin(gpu_lib)
