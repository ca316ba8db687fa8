"""Config 1 of BASELINE.json ("reference plumbing, no GPU") and SURVEY.md §8f N3: the reference's own, unmodified
``mimic3_tts`` package (imported from /root/reference) synthesises through the MI355X engine's ``onnxruntime`` shim.

Only runs where the reference checkout exists: ``$MIMIC3_REFERENCE_DIR`` (default /root/reference — this container;
the GPU box has no copy, and the reference's sources are never vendored into this repository, so there these tests
skip with that reason).  Third-party packages the reference imports but that are not installed here are replaced by
the test-only stand-ins in tests/refshim/.  Two engines behind the shim: the CPU model of the kernels (tests/emu, same
sources as the product library) in the ``-m "not gpu"`` suite, and — ``-m gpu`` variant at the bottom — the HIP library
itself on an MI355X, for whichever machine has both a GPU and the reference tree.
"""
import importlib
import io
import json
import os
import sys
import wave

import numpy as np
import pytest

from mimic3_amd import _native
from mimic3_amd import weights as W
from mimic3_amd.config import VitsConfig
from oracle.vits_oracle import VitsOracle, audio_float_to_int16
from tests.onnx_fixture import export_onnx

REFERENCE = os.environ.get("MIMIC3_REFERENCE_DIR", "/root/reference")
REFSHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")
SYMBOLS = ["_", "#", "a", "b", "c", "d", "e", "f", "g", "h", "i", "k", "l", "m", "n", "o", "r", "s", "t", "u"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "mimic3_tts")),
                                reason="no reference checkout here (set MIMIC3_REFERENCE_DIR; the GPU lease only receives "
                                       "this repository and the reference's sources are never copied into it)")


def _write_voice(root, cfg, weights, name="tiny_low", lang="en_UK", speakers=None):
    d = root / lang / name
    d.mkdir(parents=True)
    (d / "generator.onnx").write_bytes(export_onnx(cfg, weights, weight_norm_prefixes=("flow.", "dec.")))
    conf = json.loads(cfg.to_json())
    conf["inference"] = {"length_scale": 1.0, "noise_scale": 0.0, "noise_w": 0.0}
    conf["phonemizer"] = "symbols"
    conf["text_language"] = "en"
    conf["phonemes"] = {"word_separator": " ", "blank": "#", "pad": "_", "blank_between": "words"}
    (d / "config.json").write_text(json.dumps(conf))
    (d / "phonemes.txt").write_text("".join(f"{i} {s}\n" for i, s in enumerate(SYMBOLS)))
    if speakers:
        (d / "speakers.txt").write_text("".join(s + "\n" for s in speakers))
        (d / "speaker_map.csv").write_text("".join(f"{i}|tiny|{s}\n" for i, s in enumerate(speakers)))
    return d


def _import_reference(monkeypatch, tmp_path, library):
    """Import the unmodified reference with the shim registered as `onnxruntime` (INTEGRATION.md option A);
    ``library``: the CPU model of the kernels, or None for the product's own HIP library."""
    monkeypatch.syspath_prepend(REFSHIM)
    monkeypatch.syspath_prepend(REFERENCE)
    monkeypatch.setenv("XDG_DATA_HOME", str(tmp_path / "xdg"))
    monkeypatch.setenv("XDG_DATA_DIRS", str(tmp_path / "xdg-dirs"))
    if library is not None:
        monkeypatch.setattr(_native, "default_library", lambda: library)  # no GPU in this container
    import mimic3_amd.onnxruntime_shim as shim

    monkeypatch.setitem(sys.modules, "onnxruntime", shim.install())
    before = set(sys.modules)
    mimic3_tts = importlib.import_module("mimic3_tts")

    def cleanup():
        mimic3_tts.voice.Mimic3Voice._SHARED_MODELS.clear()
        for m in set(sys.modules) - before:
            if m.split(".")[0] in ("mimic3_tts", "opentts_abc", "dataclasses_json", "xdgenvpy", "gruut_ipa", "phonemes2ids",
                                   "gruut", "epitran", "espeak_phonemizer"):
                sys.modules.pop(m, None)

    return mimic3_tts, cleanup


@pytest.fixture
def reference(emu_lib, monkeypatch, tmp_path):
    mod, cleanup = _import_reference(monkeypatch, tmp_path, emu_lib)
    yield mod
    cleanup()


@pytest.fixture
def reference_on_gpu(monkeypatch, tmp_path):
    mod, cleanup = _import_reference(monkeypatch, tmp_path, None)
    yield mod
    cleanup()


def _expected_ids(text):
    ids = [1]  # blank at start
    words = text.split(" ")
    for wi, wd in enumerate(words):
        ids += [SYMBOLS.index(ch) for ch in wd]
        if wi < len(words) - 1:
            ids.append(1)
    return ids + [1]


def test_unmodified_reference_speaks_through_the_engine(reference, tmp_path):
    _speaks_through_the_engine(reference, tmp_path)


@pytest.mark.gpu
def test_unmodified_reference_speaks_through_the_hip_library(reference_on_gpu, tmp_path):
    """N3 on the device: the same unmodified ``mimic3_tts`` run, the session behind it being libmi355vits.so on an
    MI355X (needs a machine with both a GPU and the reference checkout: MIMIC3_REFERENCE_DIR)."""
    from mimic3_amd._native import default_library

    assert "gfx950" in default_library().version()
    _speaks_through_the_engine(reference_on_gpu, tmp_path)


def _speaks_through_the_engine(reference, tmp_path):
    cfg = VitsConfig.tiny()
    assert cfg.num_symbols == len(SYMBOLS)
    w = W.synthetic_weights(cfg, seed=21, frames_per_id=3.0)
    _write_voice(tmp_path / "voices", cfg, w)
    settings = reference.Mimic3Settings(voice="en_UK/tiny_low", voices_directories=[tmp_path / "voices"], no_download=True)
    tts = reference.Mimic3TextToSpeechSystem(settings)
    found = {v.key: v for v in tts.get_voices()}  # local voices first, then the downloadable catalogue (voices.json)
    assert "en_UK/tiny_low" in found and found["en_UK/tiny_low"].location == str((tmp_path / "voices/en_UK/tiny_low").absolute())

    text = "said the gull"
    tts.begin_utterance()
    tts.speak_text(text)
    results = list(tts.end_utterance())
    assert len(results) == 1
    res = results[0]
    assert (res.sample_rate_hz, res.sample_width_bytes, res.num_channels) == (22050, 2, 1)
    got = np.frombuffer(res.audio_bytes, dtype=np.int16)

    # the session the reference built is ours, loaded from generator.onnx, shared through its own cache
    voice = tts._loaded_voices["en_UK/tiny_low"]
    assert type(voice).__name__ == "SymbolsVoice"
    assert type(voice.onnx_model).__module__ == "mimic3_amd.session"
    assert list(reference.voice.Mimic3Voice._SHARED_MODELS) == [str((tmp_path / "voices/en_UK/tiny_low/generator.onnx").absolute())]

    # same ids through the oracle + the reference's own int16 conversion (voice.py:229-232)
    ids = _expected_ids(text)
    assert list(voice.phonemes_to_ids(next(iter(voice.text_to_phonemes(text)))[0])) == ids
    ora = VitsOracle(cfg, w).infer(np.array([ids]), np.array([len(ids)]), np.array([0.0, 1.0, 0.0], np.float32))
    want = audio_float_to_int16(ora["audio"][0, 0, : int(ora["audio_lengths"][0])])
    assert got.shape == want.shape and got.size > 0
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 2  # fp32 engine vs fp32 oracle, in int16 LSBs
    assert np.mean(got != want) < 0.10  # the reference's own criterion (tests/samples_match.py): <= 10 % differ

    # rate / length_scale travel through `scales` (voice.py:169-189): slower speech -> more samples
    tts.settings.rate = 0.5
    tts.begin_utterance()
    tts.speak_text(text)
    slow = np.frombuffer(list(tts.end_utterance())[0].audio_bytes, dtype=np.int16)
    assert slow.size > got.size

    # text_to_wav (opentts_abc/__init__.py) assembles a RIFF file around the same bytes
    tts.settings.rate = 1.0
    wav_bytes = tts.text_to_wav(text)
    with wave.open(io.BytesIO(wav_bytes), "rb") as wf:
        assert (wf.getframerate(), wf.getsampwidth(), wf.getnchannels()) == (22050, 2, 1)
        assert np.array_equal(np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16), got)


def test_unmodified_reference_multispeaker_and_volume(reference, tmp_path):
    cfg = VitsConfig.tiny(n_speakers=3)
    w = W.synthetic_weights(cfg, seed=22, frames_per_id=3.0)
    _write_voice(tmp_path / "voices", cfg, w, name="trio_low", speakers=["ann", "bob", "cy"])
    settings = reference.Mimic3Settings(voice="en_UK/trio_low", voices_directories=[tmp_path / "voices"], no_download=True,
                                        speaker="bob", volume=50.0)
    tts = reference.Mimic3TextToSpeechSystem(settings)
    tts.begin_utterance()
    tts.speak_text("dig a hole")
    half = np.frombuffer(list(tts.end_utterance())[0].audio_bytes, dtype=np.int16)
    ids = _expected_ids("dig a hole")
    ora = VitsOracle(cfg, w).infer(np.array([ids]), np.array([len(ids)]), np.array([0.0, 1.0, 0.0], np.float32), sid=np.array([1]))
    want = audio_float_to_int16(ora["audio"][0, 0, : int(ora["audio_lengths"][0])])
    assert half.shape == want.shape
    assert np.abs(half.astype(np.int32) - (want.astype(np.int32) // 2)).max() <= 2  # audioop.mul at volume 50 (tts.py:542-543)
    other = reference.Mimic3Settings(voice="en_UK/trio_low", voices_directories=[tmp_path / "voices"], no_download=True, speaker="cy")
    tts2 = reference.Mimic3TextToSpeechSystem(other)
    tts2.begin_utterance()
    tts2.speak_text("dig a hole")
    cy = np.frombuffer(list(tts2.end_utterance())[0].audio_bytes, dtype=np.int16)
    assert cy.shape != half.shape or not np.array_equal(cy // 2, half)  # a different speaker embedding was used


def test_reference_errors_surface_as_exceptions(reference, tmp_path):
    cfg = VitsConfig.tiny()
    w = W.synthetic_weights(cfg, seed=23)
    d = _write_voice(tmp_path / "voices", cfg, w)
    (d / "generator.onnx").write_bytes(b"this is not an onnx file")
    settings = reference.Mimic3Settings(voice="en_UK/tiny_low", voices_directories=[tmp_path / "voices"], no_download=True)
    tts = reference.Mimic3TextToSpeechSystem(settings)
    tts.begin_utterance()
    with pytest.raises(ValueError, match="cannot load"):  # raised from Mimic3Voice._load_model, like an ORT load failure
        tts.speak_text("bad")
        list(tts.end_utterance())
