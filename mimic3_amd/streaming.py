"""Chunked streaming of long-form audio (BASELINE.json configs[4]: "long-form SSML (~10k chars chunked)
streamed through mimic3_http").

The reference synthesises a request sentence by sentence — ``Mimic3TextToSpeechSystem.end_utterance`` yields one
``AudioResult`` per sentence (``mimic3_tts/tts.py:470-515``) — but the server only answers when the LAST sentence is done:
``text_to_wav`` joins everything into one WAV (``opentts_abc/__init__.py:117-127``, ``mimic3_http/app.py:157-227``,
``synthesis.py:64-73``).  With the engine behind it a 10k-character request is a few hundred sentences that finish in
tens of milliseconds each; what the listener waits for is the join.

``stream_sentences`` keeps ``look_ahead`` sentences in flight on ONE shared session (worker threads call ``run_pcm16``
concurrently, exactly like the server's synthesis workers; the session's lanes / micro-batcher / device round-robin turn
that into batched calls spread over the GPUs) and yields the int16 audio of each sentence IN ORDER as soon as it and all
its predecessors are done.  ``stream_wav`` frames that as one streamed RIFF/WAVE body: header first (data size unknown:
0xFFFFFFFF, which players accept for streams), then PCM chunks — the body of a chunked HTTP response
(INTEGRATION.md §7 shows the route next to the reference's ``/api/tts``).

Nothing here touches the GPU directly; it is host-side scheduling over ``InferenceSession``.
"""
from __future__ import annotations

import struct
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, Iterator, Optional, Sequence

import numpy as np

from .postprocess import silence

STREAM_SIZE = 0xFFFFFFFF  # RIFF / data chunk size of a WAV whose length is not known when the header is sent


def wav_stream_header(sample_rate: int = 22050) -> bytes:
    """44-byte PCM-16 mono RIFF header for a stream of unknown length."""
    return (b"RIFF" + struct.pack("<I", STREAM_SIZE) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16) + b"data" + struct.pack("<I", STREAM_SIZE))


def _feed(ids: Sequence[int], scales, sid: Optional[int]) -> Dict[str, np.ndarray]:
    arr = np.asarray(ids, dtype=np.int64).reshape(1, -1)  # voice.py:180
    feed = {"input": arr, "input_lengths": np.array([arr.shape[1]], np.int64), "scales": np.asarray(scales, np.float32)}
    if sid is not None:
        feed["sid"] = np.array([int(sid)], np.int64)
    return feed


def stream_sentences(session, sentences: Iterable[Sequence[int]], scales=(0.667, 1.0, 0.8), sid: Optional[int] = None,
                     volume: Optional[float] = None, look_ahead: int = 8) -> Iterator[np.ndarray]:
    """Yield the int16 audio of each sentence (a sequence of phoneme ids, the boundary the reference crosses at
    ``voice.py:180``) in order, keeping up to ``look_ahead`` sentences in flight on ``session``.

    An exception of a sentence surfaces when its turn comes (like the reference, which raises at the failing sentence);
    sentences after it are cancelled or drained, never yielded."""
    if look_ahead < 1:
        raise ValueError("look_ahead must be >= 1")
    it = iter(sentences)
    pool = ThreadPoolExecutor(max_workers=look_ahead, thread_name_prefix="mi355vits-stream")
    pending = []

    def submit_next() -> bool:
        try:
            ids = next(it)
        except StopIteration:
            return False
        pending.append(pool.submit(lambda f=_feed(ids, scales, sid): session.run_pcm16(f, volume=volume)[0][0]))
        return True

    try:
        while len(pending) < look_ahead and submit_next():
            pass
        while pending:
            fut = pending.pop(0)
            audio = fut.result()  # raises here if this sentence failed
            submit_next()         # keep the window full while the consumer handles this chunk
            yield audio
    finally:
        for f in pending:
            f.cancel()
        pool.shutdown(wait=True)


def stream_wav(session, sentences: Iterable[Sequence[int]], sample_rate: int = 22050, break_ms: Optional[float] = None,
               **kw) -> Iterator[bytes]:
    """The same as one streamed WAV body: header, then one chunk of little-endian PCM per sentence (with an optional
    ``add_break``-style pause between sentences, ``tts.py:452-465``)."""
    yield wav_stream_header(sample_rate)
    first = True
    for audio in stream_sentences(session, sentences, **kw):
        if not first and break_ms:
            yield silence(break_ms, sample_rate).astype("<i2").tobytes()
        first = False
        yield np.ascontiguousarray(audio, dtype="<i2").tobytes()
