"""``/api/tts/stream`` next to the reference's ``/api/tts`` (BASELINE.json configs[4]: "long-form SSML
(~10k chars chunked) streamed through mimic3_http").

The unchanged server answers a request only after its LAST sentence: ``do_synthesis`` joins every ``AudioResult`` of
``SSMLSpeaker.speak`` / ``end_utterance`` into one WAV (``mimic3_http/synthesis.py:36-85``, ``app.py:157-227``), and the sentences
of a request run one after the other on one worker thread (``tts.py:470-515``).  With the engine behind ``onnx_model.run`` a
sentence takes a few milliseconds; what a listener waits for is the join.

This module streams the same bytes sentence by sentence, with ``look_ahead`` sentences in flight on the shared session, using
ONLY the reference's own objects:

1. *plan*: the request goes through the reference's unmodified front end (``SSMLSpeaker`` or ``speak_text`` →
   ``end_utterance``) with ``Mimic3TextToSpeechSystem._speak_sentence_phonemes`` (``tts.py:519-551``, the per-sentence unit: ids,
   ``ids_to_audio``, volume) in *recording* mode for this thread only: every sentence yields an empty placeholder and its
   ``(phonemes, settings)`` are noted; breaks (``add_break``, zero bytes made on the host by the reference itself,
   ``tts.py:452-465``) and marks pass through.  Text processing only: microseconds per sentence.
2. *stream*: the recorded sentences are handed to the ORIGINAL ``_speak_sentence_phonemes`` on a small thread pool — the same
   calls the server's own workers make, so the session's lanes / micro-batcher / device round-robin batch them — and the
   results are emitted in order: RIFF header for a stream of unknown length first, then PCM per sentence (and the breaks where
   the reference put them).

The concatenated PCM equals what ``/api/tts`` returns for the same request (tested in tests/test_reference_http.py on the CPU
model of the kernels).  Nothing here touches the GPU; it is scheduling above ``InferenceSession``.
"""
from __future__ import annotations

import asyncio
import threading
from concurrent.futures import ThreadPoolExecutor
from copy import deepcopy
from typing import Any, Iterator, List, Optional, Tuple

from .streaming import wav_stream_header

_recording = threading.local()
_patch_lock = threading.Lock()


def _install_recorder(tts_cls) -> None:
    """Wrap ``tts_cls._speak_sentence_phonemes`` once: transparent unless the calling thread is planning a stream."""
    with _patch_lock:
        if getattr(tts_cls, "_mi355_stream_recorder", False):
            return
        original = tts_cls._speak_sentence_phonemes

        def _speak_sentence_phonemes(self, sent_phonemes, settings=None):
            plan = getattr(_recording, "plan", None)
            if plan is None:
                return original(self, sent_phonemes, settings=settings)
            settings = deepcopy(settings or self.settings)
            if not settings.voice:
                settings.voice = self.voice  # resolved NOW: a later request may change tts.voice while this one still streams
            plan.append((list(sent_phonemes), settings))  # (the caller clears its phoneme list right after the call)
            from opentts_abc import AudioResult

            # the sample rate is the VOICE's, as in the original (tts.py:545-551); loading is what the original would do next
            voice = self._get_or_load_voice(settings.voice or self.voice)
            ph = AudioResult(sample_rate_hz=voice.config.audio.sample_rate, audio_bytes=b"", sample_width_bytes=2, num_channels=1)
            ph._mi355_placeholder = True  # told apart from a zero-length break by this mark, not by emptiness
            return ph

        tts_cls._speak_sentence_phonemes = _speak_sentence_phonemes
        tts_cls._mi355_speak_sentence_original = original
        tts_cls._mi355_stream_recorder = True


def plan_request(tts, text: str, ssml: bool = False, text_language: Optional[str] = None) -> List[Tuple[str, Any]]:
    """The request as the reference's front end splits it: ``[("rate", hz), ("speak", (phonemes, settings)) | ("bytes", pcm_bytes),
    ...]`` — the first entry is the sample rate of the first audio result (what ``do_synthesis`` puts in the WAV header,
    ``synthesis.py:64-69``), absent when the request produces no audio at all."""
    _install_recorder(type(tts))
    from opentts_abc import AudioResult

    sentences: list = []
    _recording.plan = sentences
    try:
        if ssml:
            from opentts_abc.ssml import SSMLSpeaker

            results = list(SSMLSpeaker(tts).speak(text))
        else:
            tts.begin_utterance()
            tts.speak_text(text, text_language=text_language)
            results = list(tts.end_utterance())
    finally:
        _recording.plan = None
    plan: List[Tuple[str, Any]] = []
    it = iter(sentences)
    for r in results:
        if isinstance(r, AudioResult):
            if not plan:
                plan.append(("rate", int(r.sample_rate_hz)))
            if getattr(r, "_mi355_placeholder", False):
                plan.append(("speak", next(it)))
            elif r.audio_bytes:
                plan.append(("bytes", r.audio_bytes))  # a break: zero samples from add_break (an empty one adds nothing)
    return plan


_POOL: Optional[ThreadPoolExecutor] = None
_POOL_WORKERS = 0
_pools_lock = threading.Lock()
_streams_lock = threading.Lock()
_ACTIVE_STREAMS = 0


def _shared_pool(look_ahead: int) -> ThreadPoolExecutor:
    """ONE pool for the whole process, created by the first stream and never replaced (a pool per request costs thread start-up on
    every call and puts no bound on the threads a burst of requests creates; a pool per look-ahead value never shrinks): two full
    windows of the first request's look-ahead, at least 32 workers.  A later request that asks for a wider window than the pool
    has workers simply gets the pool's width (``_fair_share``)."""
    global _POOL, _POOL_WORKERS
    with _pools_lock:
        if _POOL is None:
            _POOL_WORKERS = max(32, 2 * look_ahead)
            _POOL = ThreadPoolExecutor(max_workers=_POOL_WORKERS, thread_name_prefix="mi355vits-http-stream")
        return _POOL


def _fair_share(look_ahead: int) -> int:
    """Sentences one stream may keep in flight right now: the pool's workers divided among the streams that are running, never
    more than its own look-ahead, and — when the look-ahead allows it — at least two (head + one behind it), so a request that
    arrives while others stream finds free workers for its first sentences instead of queueing FIFO behind the earlier
    requests' whole windows."""
    with _streams_lock:
        n = max(1, _ACTIVE_STREAMS)
    return max(min(2, look_ahead), min(look_ahead, _POOL_WORKERS // n))


def stream_plan(tts, plan: List[Tuple[str, Any]], look_ahead: int = 16, sample_rate: Optional[int] = None) -> Iterator[bytes]:
    """WAV stream of a planned request: header, then one chunk per sentence / break, in order, ``look_ahead`` sentences in
    flight.  Touches no per-request state of ``tts`` (every sentence carries its own deep-copied settings), so any number of
    these generators may run at once on one ``tts`` object."""
    if look_ahead < 1:
        raise ValueError("look_ahead must be >= 1")
    original = type(tts)._mi355_speak_sentence_original
    rate = next((v for k, v in plan if k == "rate"), None)
    yield wav_stream_header(sample_rate or rate or tts.settings.sample_rate)
    pool = _shared_pool(look_ahead)
    pending: list = []
    todo = iter([e for e in plan if e[0] != "rate"])

    def submit_next() -> bool:
        try:
            kind, payload = next(todo)
        except StopIteration:
            return False
        if kind == "bytes":
            pending.append(payload)
        else:
            phonemes, settings = payload
            pending.append(pool.submit(lambda p=phonemes, s=settings: original(tts, p, settings=s).audio_bytes))
        return True

    global _ACTIVE_STREAMS
    with _streams_lock:
        _ACTIVE_STREAMS += 1
    try:
        while len(pending) < _fair_share(look_ahead) and submit_next():
            pass
        while pending:
            head = pending.pop(0)
            chunk = head if isinstance(head, (bytes, bytearray)) else head.result()  # raises here if this sentence failed
            while len(pending) < _fair_share(look_ahead) and submit_next():  # (the share grows back when other streams end)
                pass
            if chunk:
                yield chunk
    finally:
        with _streams_lock:
            _ACTIVE_STREAMS -= 1
        # an abandoned stream (client gone, a sentence failed): what has not started is cancelled; what is running finishes on
        # its lane (a synthesis call cannot be interrupted) and its result is dropped — at most `look_ahead` calls, and the
        # fair share above keeps them from starving the streams that go on
        for f in pending:
            if not isinstance(f, (bytes, bytearray)):
                f.cancel()


def stream_request(tts, text: str, ssml: bool = False, text_language: Optional[str] = None, look_ahead: int = 16,
                   sample_rate: Optional[int] = None) -> Iterator[bytes]:
    """plan + stream in one call (single caller: the plan step reads ``tts.settings`` / ``tts.voice``)."""
    if look_ahead < 1:
        raise ValueError("look_ahead must be >= 1")
    yield from stream_plan(tts, plan_request(tts, text, ssml=ssml, text_language=text_language), look_ahead, sample_rate)


def add_stream_route(app, tts, quart_module, args=None, look_ahead: int = 16, rule: str = "/api/tts/stream"):
    """Register ``rule`` on the Quart app ``mimic3_http.app.get_app`` returned.  ``tts``: a ``Mimic3TextToSpeechSystem`` of the
    server process (its voices share the sessions of the synthesis workers through ``Mimic3Voice._SHARED_MODELS``,
    ``voice.py:277-292``); ``quart_module``: the imported ``quart``.  Query parameters as ``/api/tts`` (``app.py:157-219``):
    ``voice``, ``noiseScale``, ``noiseW``, ``lengthScale``, ``ssml``, ``textLanguage``; text in the POST body or ``?text=``."""
    request, Response = quart_module.request, quart_module.Response
    lock = threading.Lock()  # one request at a time PLANS on this tts object (its settings / voice are per-request state)
    try:
        from mimic3_tts import DEFAULT_VOICE
    except ImportError:  # pragma: no cover - the route only exists next to the reference
        DEFAULT_VOICE = "en_UK/apope_low"

    @app.route(rule, methods=["GET", "POST"])
    async def app_tts_stream():
        a = request.args
        if request.method == "POST":
            text = (await request.data).decode()
        else:
            text = a.get("text", "")
        assert text, "No text provided"
        if args is not None and getattr(args, "max_text_length", None) is not None:
            text = text[: args.max_text_length]
        ssml_str = a.get("ssml")
        ssml = (ssml_str.strip().lower() in {"true", "1", "yes", "on"}) if ssml_str else request.content_type == "application/ssml+xml"
        voice = a.get("voice") or (getattr(args, "voice", None) if args is not None else None) or DEFAULT_VOICE  # app.py:168
        # the plan step is the only part that touches tts.settings / tts.voice: done under the lock, exactly as do_synthesis sets
        # them (synthesis.py:41-47: ALL of them on every request, None = the voice's own default), so that nothing leaks from one
        # request into the next.  It phonemises the whole text and may load a voice: seconds for a 10k-character request — so it
        # runs in a worker thread (the lock is taken THERE: a threading.Lock held across an await would stall the loop), and the
        # event loop keeps serving the other requests' chunks meanwhile.  The stream itself runs outside the lock.
        params = {name: (float(a.get(key)) if a.get(key) else (getattr(args, name, None) if args is not None else None))
                  for key, name in (("noiseScale", "noise_scale"), ("noiseW", "noise_w"), ("lengthScale", "length_scale"))}
        text_language = a.get("textLanguage")

        def locked_plan():
            with lock:
                tts.speaker = None
                tts.voice = str(voice)
                for name, v in params.items():
                    setattr(tts.settings, name, v)
                return plan_request(tts, text, ssml=ssml, text_language=text_language)

        plan = await asyncio.get_running_loop().run_in_executor(None, locked_plan)
        return Response(stream_plan(tts, plan, look_ahead=look_ahead), mimetype="audio/wav")

    return app_tts_stream
