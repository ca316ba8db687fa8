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


def _tx_class(n: int) -> int:
    """The phoneme-length classes within which a batched row gets the kernels — and the bits — it gets alone (the text encoder
    picks its attention / FFN kernels by the padded length; session._MicroBatcher groups arrivals the same way)."""
    return 0 if n <= 128 else (1 if n <= 256 else (2 if n <= 512 else 3))


def plan_batches(lengths: Sequence[int], head: int = 0, max_batch: int = 48) -> list:
    """Batches for a request whose sentences are all known (``end_utterance`` holds every pending ``Mimic3Phonemes`` when it
    starts, ``mimic3_tts/tts.py:470-515``): sentence 0 ALONE (its audio is what the listener waits for), then — optionally — the
    next ``head`` sentences as one batch (for playback that must continue before the rest is done; at 16,000 x real time the rest
    of a 10k-character request is done 15 ms later, so the default is 0), then everything else sorted by length inside its
    phoneme-length class and cut into batches of at most ``max_batch`` — rows of one batch have similar lengths, so little of a
    batched call is padding (the decoder computes every row up to the batch's longest).  Returns lists of sentence indices, in
    the order the batches should be issued (by the earliest sentence they hold)."""
    n = len(lengths)
    if n == 0:
        return []
    batches = [[0]]
    if n > 1:
        first = list(range(1, min(n, 1 + max(0, head))))
        by_class: dict = {}
        for i in first:
            by_class.setdefault(_tx_class(int(lengths[i])), []).append(i)
        batches.extend(by_class.values())
    rest = list(range(1 + max(0, head), n))
    by_class = {}
    for i in rest:
        by_class.setdefault(_tx_class(int(lengths[i])), []).append(i)
    tail = []
    for idx in by_class.values():
        idx.sort(key=lambda i: (int(lengths[i]), i))
        nb = -(-len(idx) // max(1, max_batch))
        size = -(-len(idx) // nb)  # equal shares instead of full batches + a remainder
        for k in range(0, len(idx), size):
            tail.append(sorted(idx[k:k + size]))
    tail.sort(key=lambda b: b[0])
    return batches + tail


def _batch_feed(rows: Sequence[Sequence[int]], scales, sid: Optional[int]) -> Dict[str, np.ndarray]:
    tx = max(len(r) for r in rows)
    ids = np.zeros((len(rows), tx), np.int64)
    for b, r in enumerate(rows):
        ids[b, : len(r)] = np.asarray(r, np.int64)
    feed = {"input": ids, "input_lengths": np.array([len(r) for r in rows], np.int64), "scales": np.asarray(scales, np.float32)}
    if sid is not None:
        feed["sid"] = np.full(len(rows), int(sid), np.int64)
    return feed


def stream_planned(session, sentences: Sequence[Sequence[int]], scales=(0.667, 1.0, 0.8), sid: Optional[int] = None,
                   volume: Optional[float] = None, head: int = 0, max_batch: int = 48, workers: int = 4,
                   stats: Optional[dict] = None, first_alone: bool = True) -> Iterator[np.ndarray]:
    """``stream_sentences`` for a request whose sentence list is known up front ("look-ahead batching of all
    Mimic3Phonemes pending in one end_utterance"): the batches of ``plan_batches`` are issued as BATCHED engine calls on the
    session's lanes (no arrival window, no per-sentence thread), the chunks are yielded in sentence order as their batches
    finish.  A batched row is bitwise its single call (same phoneme-length class), so the stream's bytes do not depend on the
    plan.  ``first_alone``: sentence 0 has the device to itself — the other batches are issued when its audio is in the caller's
    hands (issued together, a 40-row batch's kernels fill the chip and the single sentence takes 6 ms instead of 2:
    profiles/r06_serve_bench.log).  ``stats`` (optional dict) receives ``batches``, ``padding_efficiency`` = valid / computed
    output samples (a batch computes every row up to its longest) and ``text_padding_efficiency`` (the same for phoneme positions).

    A failing sentence: its batch is retried row by row, the error surfaces at that sentence's turn, later ones are not
    delivered."""
    sentences = list(sentences)  # (shallow: the rows are read when their batch's feed is built)
    if not sentences:
        return
    pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="mi355vits-plan")
    done: Dict[int, object] = {}
    valid = computed = tvalid = tcomputed = 0

    def prepare():
        lens = [len(s) for s in sentences]
        return plan_batches(lens, head=head, max_batch=max_batch), lens

    def run_batch(idx):
        try:
            rows, lengths = session.run_pcm16(_batch_feed([sentences[i] for i in idx], scales, sid), volume=volume)
            return list(rows), np.asarray(lengths)  # (views of the call's own pinned result buffer, which they keep alive)
        except Exception:
            if len(idx) == 1:
                raise
            out = []  # a bad row must not take its batch-mates with it: one call per row, each with its own outcome
            for i in idx:
                try:
                    out.append(session.run_pcm16(_feed(sentences[i], scales, sid), volume=volume)[0][0])
                except Exception as e:  # noqa: BLE001 - delivered at the sentence's turn
                    out.append(e)
            return out, None

    if first_alone:
        # sentence 0 in the caller's own thread, straight onto a lane: no pool hand-over, no micro-batcher queue, and the plan of the
        # rest is made on a pool thread WHILE it runs (the engine call releases the GIL) — nothing but the call itself stands between
        # the request and the audio the listener is waiting for
        from concurrent.futures import Future

        prep = pool.submit(prepare)
        f0: Future = Future()
        try:
            try:
                r0 = session.run_pcm16(_feed(sentences[0], scales, sid), volume=volume, direct=True)
            except TypeError:  # a session-like object without the `direct` extension
                r0 = session.run_pcm16(_feed(sentences[0], scales, sid), volume=volume)
            f0.set_result((list(r0[0]), None))
        except Exception as e:  # noqa: BLE001 - raised at the sentence's turn below
            f0.set_exception(e)
        plan, lens = prep.result()
        futs = [(plan[0], f0)]
    else:
        plan, lens = prepare()
        futs = [(idx, pool.submit(run_batch, idx)) for idx in plan]
    try:
        nxt = 0
        k = 0
        while k < len(futs):
            idx, fut = futs[k]
            k += 1
            rows, lengths = fut.result()
            if first_alone and k == 1:
                futs += [(i2, pool.submit(run_batch, i2)) for i2 in plan[1:]]  # the rest starts as sentence 0 is handed over
            if lengths is not None:
                valid += int(np.sum(lengths))
                computed += int(np.max(lengths)) * len(idx)
                tvalid += sum(lens[i] for i in idx)
                tcomputed += max(lens[i] for i in idx) * len(idx)
            for i, r in zip(idx, rows):
                done[i] = r
            while nxt in done:
                r = done.pop(nxt)
                if isinstance(r, Exception):
                    raise r
                nxt += 1
                yield r
    finally:
        for _idx, f in futs:
            f.cancel()
        pool.shutdown(wait=True)
        if stats is not None:
            stats.update({"batches": [len(i) for i in plan], "padding_efficiency": valid / computed if computed else None,
                          "text_padding_efficiency": tvalid / tcomputed if tcomputed else None})


def stream_sentences(session, sentences: Iterable[Sequence[int]], scales=(0.667, 1.0, 0.8), sid: Optional[int] = None,
                     volume: Optional[float] = None, look_ahead: int = 8, plan: Optional[bool] = None,
                     stats: Optional[dict] = None) -> Iterator[np.ndarray]:
    """Yield the int16 audio of each sentence (a sequence of phoneme ids, the boundary the reference crosses at
    ``voice.py:180``) in order, keeping up to ``look_ahead`` sentences in flight on ``session``.

    ``plan``: a request that arrives as a list / tuple is planned (``stream_planned``: sentence 0 alone, then length-sorted
    batches cut from the whole list); a lazy iterable is consumed ``look_ahead`` sentences
    ahead of the consumer, one call per sentence, and batching is left to the session's micro-batcher.  Same chunks either way.

    An exception of a sentence surfaces when its turn comes (like the reference, which raises at the failing sentence);
    sentences after it are cancelled or drained, never yielded."""
    if look_ahead < 1:
        raise ValueError("look_ahead must be >= 1")
    if plan is None:
        plan = isinstance(sentences, (list, tuple))
    if plan:
        lanes = len(getattr(session, "_engines", [None]))
        yield from stream_planned(session, list(sentences), scales=scales, sid=sid, volume=volume, workers=max(2, lanes + 1), stats=stats)
        return
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
