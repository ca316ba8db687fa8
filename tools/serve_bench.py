#!/usr/bin/env python3
"""Serving-shaped measurement of the session shim: C concurrent clients, each doing what a
mimic3_http synthesis worker does — one single-utterance ``run`` (+ int16) per sentence on a session shared with the
other workers (mimic3_tts/voice.py:277-292, mimic3_http/synthesis.py:88-136) — against

  plain        one engine handle, calls serialise (what a drop-in without N2 would give)
  lanes        SessionOptions.lanes = 3
  batch        micro-batching window 2 ms, max 32
  batch+lanes  both

and, with STREAM=1, a long-form request (BASELINE.json configs[4]: ~10k characters = 120 sentences) delivered as a chunked
stream (mimic3_amd.streaming): time to first audio and total time vs synthesising everything before answering.

Prints one JSON line per mode: sentences/s, audio seconds per second, latency percentiles.  Run on the GPU box.
MI355VITS_DEVICES=all spreads every session over all visible GPUs (in-process device round-robin).
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mimic3_amd import weights as W  # noqa: E402
from mimic3_amd.config import VitsConfig  # noqa: E402
from mimic3_amd.session import InferenceSession, SessionOptions  # noqa: E402


def main():
    clients = int(os.environ.get("CLIENTS", "64"))
    seconds = float(os.environ.get("SECONDS", "4"))
    cfg = VitsConfig.apope_low()
    blob = W.pack(cfg, W.synthetic_weights(cfg, seed=1234, frames_per_id=6.0))
    rng = np.random.default_rng(0)
    # sentences of 40..160 phoneme ids (the golden utterance has ~180)
    feeds = []
    for _ in range(256):
        n = int(rng.integers(40, 161))
        feeds.append({"input": rng.integers(1, 50, (1, n)).astype(np.int64), "input_lengths": np.array([n], np.int64),
                      "scales": np.array([0.667, 1.0, 0.8], np.float32)})
    modes = {"plain": (1, 0.0), "lanes": (3, 0.0), "batch": (1, 2.0), "batch+lanes": (3, 2.0)}
    maxb = int(os.environ.get("MAXB", "32"))
    if os.environ.get("MODES"):  # e.g. MODES="batch+lanes:2:1.0,batch+lanes:4:2.0" = name:lanes:window_ms (a sweep of the serving knobs)
        modes = {}
        for k, spec in enumerate(os.environ["MODES"].split(",")):
            nm, ln, win = spec.split(":")
            modes["%s[lanes=%s,window=%s,max=%d]#%d" % (nm, ln, win, maxb, k)] = (int(ln), float(win))
    for name, (lanes, window) in modes.items():
        so = SessionOptions()
        so.lanes = lanes
        so.micro_batch_window_ms = window
        so.micro_batch_max = maxb
        sess = InferenceSession(blob, sess_options=so)
        for f in feeds[:4]:
            sess.run_pcm16(f)
        lat, samples = [], [0]
        lock = threading.Lock()
        stop = time.perf_counter() + seconds

        def client(k):
            i = k
            mine, n = [], 0
            while time.perf_counter() < stop:
                t0 = time.perf_counter()
                rows, lengths = sess.run_pcm16(feeds[i % len(feeds)])
                mine.append(time.perf_counter() - t0)
                n += int(lengths[0])
                i += clients
            with lock:
                lat.extend(mine)
                samples[0] += n

        t0 = time.perf_counter()
        ts = [threading.Thread(target=client, args=(k,)) for k in range(clients)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        wall = time.perf_counter() - t0
        lat_ms = np.sort(np.array(lat)) * 1e3
        out = {"mode": name, "clients": clients, "lanes": lanes, "micro_batch_window_ms": window, "sentences_per_s": len(lat) / wall,
               "audio_s_per_s": samples[0] / 22050 / wall, "latency_ms_p50": float(lat_ms[len(lat_ms) // 2]),
               "latency_ms_p95": float(lat_ms[int(0.95 * len(lat_ms))]), "latency_ms_max": float(lat_ms[-1])}
        if sess._batcher is not None:
            out["mean_batch"] = sess._batcher.requests / max(1, sess._batcher.batches)
        print(json.dumps(out), flush=True)
        sess.close()
    if os.environ.get("STREAM"):
        from mimic3_amd import streaming as ST

        so = SessionOptions()
        so.lanes = 3
        so.micro_batch_window_ms = 1.0
        sess = InferenceSession(blob, sess_options=so)
        sentences = [f["input"][0].tolist() for f in feeds[:120]]  # ~100 phoneme ids each: about 10k characters of text
        list(ST.stream_sentences(sess, sentences[:8], look_ahead=8))  # warm-up
        for look, planned in ((1, False), (8, False), (32, False), (32, True), (32, True)):
            t0 = time.perf_counter()
            first, n = None, 0
            stats = {}
            for audio in ST.stream_sentences(sess, sentences if planned else iter(sentences), look_ahead=look, stats=stats):
                if first is None:
                    first = time.perf_counter() - t0
                n += audio.shape[0]
            total = time.perf_counter() - t0
            print(json.dumps({"mode": "stream", "planned": planned, "sentences": len(sentences), "look_ahead": look, "first_audio_ms": first * 1e3,
                              "total_ms": total * 1e3, "audio_s": n / 22050, "x_realtime": n / 22050 / total,
                              "devices": sess.devices, **stats}), flush=True)
        sess.close()


if __name__ == "__main__":
    main()
