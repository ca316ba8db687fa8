"""BASELINE.json configs[4] / SURVEY.md §8f N4 through the reference's own server objects (CPU model of the kernels; skips
where no reference checkout exists): the unmodified ``mimic3_http.app.get_app`` + ``mimic3_http.synthesis.do_synthesis_proc``
worker threads answer a ~10k-character SSML request through the MI355X engine's ``onnxruntime`` shim — one shared session over
two (emulated) devices with lanes and micro-batching, configured by environment exactly as INTEGRATION.md §6 tells an operator —
and ``/api/tts/stream`` (mimic3_amd/http_stream.py) streams the same bytes sentence by sentence.

quart / quart_cors / swagger_ui are not installed here: tests/refshim/ holds test-only stand-ins for the few names the
reference's app touches (hypercorn is only imported by ``mimic3_http.__main__``, whose three lines of thread start-up the test
repeats)."""
import argparse
import asyncio
import importlib
import io
import os
import sys
import threading
import wave
from queue import Queue

import numpy as np
import pytest

from mimic3_amd import weights as W
from mimic3_amd.config import VitsConfig
from tests.test_reference_end_to_end import REFERENCE, SYMBOLS, _expected_ids, _import_reference, _write_voice

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "mimic3_http")),
                                reason="no reference checkout here (set MIMIC3_REFERENCE_DIR)")

WORDS = ["said", "the", "gull", "a", "big", "red", "hen", "dug", "in", "cold", "mud", "and", "faint", "stars", "rose", "on", "high"]


def _sentences(n, rng):
    return [" ".join(rng.choice(WORDS, size=int(rng.integers(8, 15)))) for _ in range(n)]


@pytest.fixture
def server(emu_lib, monkeypatch, tmp_path):
    """The reference's app + 3 synthesis workers over one voice; the session behind them: 2 devices x 2 lanes, 2 ms micro-batching."""
    monkeypatch.setenv("MI355_EMU_DEVICES", "2")
    monkeypatch.setenv("MI355VITS_DEVICES", "all")
    monkeypatch.setenv("MI355VITS_LANES", "2")
    monkeypatch.setenv("MI355VITS_MICROBATCH_MS", "2")
    mimic3_tts, cleanup = _import_reference(monkeypatch, tmp_path, emu_lib)
    before = set(sys.modules)
    app_mod = importlib.import_module("mimic3_http.app")
    syn_mod = importlib.import_module("mimic3_http.synthesis")
    from mimic3_http.args import _MISSING

    cfg = VitsConfig.tiny()
    assert cfg.num_symbols == len(SYMBOLS)
    w = W.synthetic_weights(cfg, seed=31, frames_per_id=2.0)
    _write_voice(tmp_path / "voices", cfg, w)
    args = argparse.Namespace(voices_dir=[str(tmp_path / "voices")], voice="en_UK/tiny_low", speaker=None, noise_scale=0.0, length_scale=1.0,
                              noise_w=0.0, cache_dir=_MISSING, preload_voice=["en_UK/tiny_low"], cuda=False, deterministic=True,
                              num_threads=3, max_text_length=None, default_voice=None, play_program="true", no_show_openapi=True,
                              debug=False)
    q: Queue = Queue()
    threads = [threading.Thread(target=syn_mod.do_synthesis_proc, args=(args, q), daemon=True) for _ in range(args.num_threads)]
    [t.start() for t in threads]
    app = app_mod.get_app(args, q, str(tmp_path / "cache"))
    yield mimic3_tts, app, args, cfg, w, emu_lib
    for _ in threads:
        q.put(None)
    [t.join(timeout=30) for t in threads]
    for m in set(sys.modules) - before:
        if m.split(".")[0] in ("mimic3_http", "quart", "quart_cors", "swagger_ui"):
            sys.modules.pop(m, None)
    cleanup()


def _pcm_of(wav_bytes):
    with wave.open(io.BytesIO(wav_bytes), "rb") as wf:
        assert (wf.getframerate(), wf.getsampwidth(), wf.getnchannels()) == (22050, 2, 1)
        return np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)


def test_unmodified_server_answers_long_form_ssml_through_the_engine_and_streams_it(server, tmp_path):
    mimic3_tts, app, args, cfg, w, lib = server
    from mimic3_amd.session import InferenceSession
    from mimic3_tts.utils import audio_float_to_int16

    rng = np.random.default_rng(17)
    sents = _sentences(200, rng)
    ssml = "<speak>" + "".join(f"<s>{s}</s>" for s in sents) + "</speak>"
    assert len(ssml) > 10000

    # ---- /api/tts of the unmodified app: request -> queue -> worker thread -> SSMLSpeaker -> one run() per sentence
    resp = asyncio.run(app.dispatch("POST", "/api/tts", args={"ssml": "true"}, body=ssml.encode()))
    assert resp.status_code == 200 and resp.mimetype == "audio/wav", asyncio.run(resp.get_data())[:300]
    got = _pcm_of(asyncio.run(resp.get_data()))

    # the session the workers share is ours, on both devices, and it did batch concurrent sentences... of ONE request there
    # is only one in flight (the reference speaks a request's sentences one after the other: tts.py:470-515)
    models = mimic3_tts.voice.Mimic3Voice._SHARED_MODELS
    assert len(models) == 1
    sess = next(iter(models.values()))
    assert type(sess).__module__ == "mimic3_amd.session" and sess.devices == [0, 1] and len(sess._engines) == 4

    # per-sentence audio of direct calls (plain single-lane session, same ids, the reference's own int16 conversion)
    plain = InferenceSession(os.path.join(str(tmp_path / "voices"), "en_UK", "tiny_low", "generator.onnx"), _library=lib)
    want = []
    for s in sents:
        ids = np.array([_expected_ids(s)], np.int64)
        a = plain.run(None, {"input": ids, "input_lengths": np.array([ids.shape[1]], np.int64),
                             "scales": np.array([0.0, 1.0, 0.0], np.float32)})[0].squeeze()
        want.append(audio_float_to_int16(a))
    want_all = np.concatenate(want)
    assert got.shape == want_all.shape and np.array_equal(got, want_all)

    # ---- /api/tts/stream: same bytes, sentence by sentence, 16 sentences in flight on the shared session
    from mimic3_amd import http_stream
    import quart

    tts = mimic3_tts.Mimic3TextToSpeechSystem(mimic3_tts.Mimic3Settings(voice=args.voice, voices_directories=args.voices_dir,
                                                                         noise_scale=0.0, noise_w=0.0, length_scale=1.0))
    http_stream.add_stream_route(app, tts, quart, args=args, look_ahead=16)
    calls_before = dict(sess._free_lanes.calls_per_device)
    sresp = asyncio.run(app.dispatch("POST", "/api/tts/stream", args={"ssml": "true"}, body=ssml.encode()))
    chunks = list(sresp.response)
    assert chunks[0][:4] == b"RIFF" and chunks[0][8:12] == b"WAVE" and len(chunks[0]) == 44
    assert len(chunks) == 1 + len(sents)
    for c, e in zip(chunks[1:], want):
        assert np.array_equal(np.frombuffer(c, dtype=np.int16), e)
    # with 16 sentences in flight the micro-batcher coalesced calls and both devices took part
    calls = {d: sess._free_lanes.calls_per_device[d] - calls_before[d] for d in calls_before}
    assert all(n > 0 for n in calls.values()) and sum(calls.values()) < len(sents), calls

    # breaks stay where the reference puts them; a plain-text request streams too
    with_break = "<speak><s>said the gull</s><break time=\"100ms\"/><s>a big red hen</s></speak>"
    a = _pcm_of(asyncio.run(asyncio.run(app.dispatch("POST", "/api/tts", args={"ssml": "true"}, body=with_break.encode())).get_data()))
    b = b"".join(list(asyncio.run(app.dispatch("POST", "/api/tts/stream", args={"ssml": "true"}, body=with_break.encode())).response)[1:])
    assert np.array_equal(a, np.frombuffer(b, dtype=np.int16)) and int(0.1 * 22050) * 2 <= len(b)

    # per-request settings do not leak into the next request (do_synthesis assigns all of them every time, synthesis.py:41-47):
    # a request with lengthScale=2 is longer, the next one WITHOUT the parameter is the default again and equals /api/tts
    two = "<speak><s>said the gull</s><s>a big red hen</s></speak>"

    def stream_pcm(extra=None, body=two):
        r = asyncio.run(app.dispatch("POST", "/api/tts/stream", args=dict({"ssml": "true"}, **(extra or {})), body=body.encode()))
        return np.frombuffer(b"".join(list(r.response)[1:]), dtype=np.int16)

    base = _pcm_of(asyncio.run(asyncio.run(app.dispatch("POST", "/api/tts", args={"ssml": "true"}, body=two.encode())).get_data()))
    slow = stream_pcm({"lengthScale": "2"})
    assert len(slow) > 1.5 * len(base)
    assert np.array_equal(stream_pcm(), base)
    assert tts.settings.length_scale == args.length_scale and tts.voice == (args.voice or mimic3_tts.DEFAULT_VOICE)
    # a zero-length break is not mistaken for a recorded sentence (it used to consume one and fail the request)
    zero = "<speak><s>said the gull</s><break time=\"0ms\"/><s>a big red hen</s></speak>"
    assert np.array_equal(stream_pcm(body=zero), base)
    # two stream requests at once: the lock covers the plan step only, both streams run and both are right
    import threading
    res = {}
    ths = [threading.Thread(target=lambda k=k: res.__setitem__(k, stream_pcm({"lengthScale": "2"} if k else None))) for k in (0, 1)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert np.array_equal(res[0], base) and np.array_equal(res[1], slow)
    plain.close()
    tts._loaded_voices.clear()


def test_server_error_path_is_the_references(server):
    """A failing request comes back through the app's own error handler (app.py:349-353): '<Class>: msg', 500."""
    mimic3_tts, app, args, cfg, w, lib = server
    resp = asyncio.run(app.dispatch("POST", "/api/tts", args={"voice": "xx_XX/nope_low"}, body=b"said the gull"))
    body = asyncio.run(resp.get_data()).decode()
    assert resp.status_code == 500 and "VoiceNotFoundError" in body, body
    ok = asyncio.run(app.dispatch("GET", "/api/healthcheck"))
    assert asyncio.run(ok.get_data()) == b"OK"


def test_fair_share_of_concurrent_streams_with_a_fake_tts():
    """the accounting of concurrent streams (one shared pool, a fair share of its workers per running stream, look_ahead = 1
    honoured, an abandoned generator gives its share back) — driven with a fake tts, no engine involved."""
    import threading
    import time

    from mimic3_amd import http_stream as HS

    class FakeTTS:
        def __init__(self):
            self.inflight = 0
            self.peak = 0
            self.lock = threading.Lock()

        def speak(self, p, settings=None):
            with self.lock:
                self.inflight += 1
                self.peak = max(self.peak, self.inflight)
            time.sleep(0.002)
            with self.lock:
                self.inflight -= 1

            class R:
                audio_bytes = bytes([p % 256]) * 4
            return R()

    FakeTTS._mi355_speak_sentence_original = lambda self, p, settings=None: self.speak(p, settings)
    tts = FakeTTS()
    tts.settings = type("S", (), {"sample_rate": 22050})()
    plan = [("rate", 22050)] + [("speak", (i, None)) for i in range(12)]
    # look_ahead = 1: never more than one sentence of this stream in flight (round 5 kept two)
    body = b"".join(HS.stream_plan(tts, plan, look_ahead=1))
    assert body[44:] == b"".join(bytes([i]) * 4 for i in range(12)) and tts.peak == 1
    assert HS._fair_share(1) == 1 and HS._ACTIVE_STREAMS == 0
    # 40 concurrent streams on the one pool: every stream complete and in order, the share never exceeds the pool
    outs = [None] * 40

    def run(k):
        outs[k] = b"".join(HS.stream_plan(tts, plan, look_ahead=8))

    ts = [threading.Thread(target=run, args=(k,)) for k in range(40)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(o == body for o in outs) and tts.peak <= HS._POOL_WORKERS and HS._ACTIVE_STREAMS == 0
    # an abandoned generator (client gone after two chunks) returns its share
    g = HS.stream_plan(tts, plan, look_ahead=4)
    next(g), next(g), next(g)
    assert HS._ACTIVE_STREAMS == 1
    g.close()
    assert HS._ACTIVE_STREAMS == 0
