"""Host-side mirror of the one interface Mimic 3 uses on its hot path.

``Mimic3Voice`` keeps an ``onnxruntime.InferenceSession`` in ``self.onnx_model`` and calls exactly
``self.onnx_model.run(None, inputs)[0].squeeze()`` (``mimic3_tts/voice.py:230``); the session is built in
``Mimic3Voice._load_model`` (``voice.py:378-407``) from ``onnxruntime.SessionOptions()`` with
``graph_optimization_level`` / ``use_deterministic_compute`` assigned and ``providers=None`` or
``["CUDAExecutionProvider"]`` (``tts.py:590-593``).  The classes below keep those names, argument
meanings and error behaviour (Python exceptions, never partial audio) and forward to the MI355X engine
through the C ABI (``include/mi355vits.h``).  There is no CPU execution path here.

Voice files: the reference passes ``<voice_dir>/generator.onnx`` (``voice.py:273``).  The engine reads the
voice from ``generator.m355`` beside it (weights + hyper-parameters, written by ``mimic3_amd.weights.save`` or
``python -m mimic3_amd.onnx_import``) when that exists and belongs to the ``.onnx`` (the ``.onnx`` is absent or an
empty placeholder, or the container's trailer records exactly this file's size and sha256); otherwise the ONNX
initialisers are converted in memory at load time.  A path that already ends in ``.m355`` is used as is.
"""
from __future__ import annotations

import itertools
import os
import threading
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _native
from .config import VitsConfig


class GraphOptimizationLevel:
    """Same member names as ``onnxruntime.GraphOptimizationLevel`` (``voice.py:397-399`` sets
    ``ORT_DISABLE_ALL`` on armv7l).  The engine has a fixed launch sequence, so the level is recorded only."""

    ORT_DISABLE_ALL = 0
    ORT_ENABLE_BASIC = 1
    ORT_ENABLE_EXTENDED = 2
    ORT_ENABLE_ALL = 99


class SessionOptions:
    """Attribute bag with the fields the reference assigns (``voice.py:392-401``)."""

    def __init__(self):
        self.graph_optimization_level = GraphOptimizationLevel.ORT_ENABLE_ALL
        self.use_deterministic_compute = False
        self.intra_op_num_threads = 0
        self.inter_op_num_threads = 0
        self.log_severity_level = 2
        # engine extensions (not in onnxruntime)
        self.device_id: Optional[int] = None
        self.seed: Optional[int] = None
        # caller-side micro-batching: concurrent single-utterance run() calls of the server's
        # worker threads that arrive within this window are synthesised as ONE batched engine call (a batch is
        # bitwise equal to separate calls).  0 = off.  Also MI355VITS_MICROBATCH_MS.
        self.micro_batch_window_ms: float = float(os.environ.get("MI355VITS_MICROBATCH_MS", "0") or 0)
        self.micro_batch_max: int = 64
        # engine handles (each with its own HIP stream and workspace) that concurrent run() calls are spread over.
        # The server's worker threads call run() on one shared session without a lock (voice.py:277-292); with one
        # lane those calls queue up behind each other, with two the launch-bound text-encoder / duration-predictor
        # half of one request overlaps the matrix-core half of another (+10 % throughput at batch 32).
        # Results do not depend on the lane.  Also MI355VITS_LANES.
        self.lanes: int = max(1, int(os.environ.get("MI355VITS_LANES", "1") or 1))
        # in-process multi-GPU (SURVEY.md §8f N2 "device round-robin"): the unchanged mimic3-server runs its
        # --num-threads synthesis workers in ONE process on ONE shared session (mimic3_http/__main__.py:53-61,
        # voice.py:277-292), so using the 8 GPUs of a node behind it means this session owns all of them.
        # A list of device indices, or "all"; `lanes` is then per device, the weights are uploaded once per device and
        # shared by its lanes, and every call goes to the least-loaded device.  None = one device (device_id /
        # provider options / MI355VITS_DEVICE / LOCAL_RANK).  Also MI355VITS_DEVICES="0,1,2,3" or "all".
        self.devices: Union[None, str, Sequence[int]] = os.environ.get("MI355VITS_DEVICES") or None
        # matrix-core path of the dense convs on every handle of this session: "bf16x3" (default, f32-grade), "f32",
        # "bf16w" (BASELINE configs[4]: bf16 weights, reduced precision, same utterance lengths), "f16x2".  None = the
        # library's own default (environment MI355VITS_MATH, read when a handle is created).  include/mi355vits.h.
        self.math: Optional[str] = None


class NodeArg:
    def __init__(self, name: str, type_: str, shape: Sequence[Any]):
        self.name = name
        self.type = type_
        self.shape = list(shape)

    def __repr__(self):
        return f"NodeArg(name='{self.name}', type='{self.type}', shape={self.shape})"


class InvalidArgument(ValueError):
    """Bad feed (what onnxruntime reports as ``InvalidArgument``)."""


def resolve_voice_file(path: Union[str, os.PathLike]) -> Union[str, bytes]:
    """What to hand the engine for the model path Mimic 3 passes (``voice.py:273,403``): the ``.m355`` container
    beside it when there is one, otherwise the ONNX file's initialisers converted in memory
    (``mimic3_amd.onnx_import``).  ``python -m mimic3_amd.onnx_import generator.onnx`` writes the
    container once so that later loads skip the conversion."""
    p = os.fspath(path)
    if p.endswith(".m355"):
        if not os.path.isfile(p):
            raise FileNotFoundError(f"{p} not found")
        return p
    cand = os.path.splitext(p)[0] + ".m355"
    if os.path.isfile(cand) and _container_belongs_to(cand, p):
        return cand
    if not os.path.isfile(p):
        raise FileNotFoundError(f"neither {p} nor {cand} found (see INTEGRATION.md)")
    from . import onnx_import
    with open(p, "rb") as f:
        blob = f.read()
    try:
        return onnx_import.onnx_to_m355_bytes(blob, onnx_import.load_voice_config(p), p)
    except onnx_import.OnnxImportError as e:
        raise InvalidArgument(f"cannot load {p}: {e}") from e


def _container_belongs_to(m355_path: str, onnx_path: str) -> bool:
    """Explicit rule (no timestamps): the container stands in for ``onnx_path`` when that file is absent or an empty
    placeholder, or when the container's trailer (``weights.source_record``) names exactly that file — same size and
    same sha256.  A re-downloaded / replaced ``.onnx`` therefore falls back to in-memory conversion."""
    if not os.path.isfile(onnx_path):
        return True
    size = os.path.getsize(onnx_path)
    if size == 0:
        return True
    from . import weights as W

    rec = W.read_source_record(m355_path)
    if rec is None or rec[0] != size:
        return False
    import hashlib

    h = hashlib.sha256()
    with open(onnx_path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.digest() == rec[1]


def _model_bytes(blob: bytes) -> bytes:
    """``InferenceSession(model_bytes)``: an ``.m355`` container as is, anything else is taken for an ONNX model."""
    if blob[:8] == b"M355VITS":
        return blob
    from . import onnx_import
    try:
        return onnx_import.onnx_to_m355_bytes(blob)
    except onnx_import.OnnxImportError as e:
        raise InvalidArgument(f"cannot load model bytes: {e}") from e


def _device_from_providers(providers, provider_options, sess_options) -> int:
    if sess_options is not None and getattr(sess_options, "device_id", None) is not None:
        return int(sess_options.device_id)
    dev = None
    if providers:
        for i, p in enumerate(providers):
            opts = None
            if isinstance(p, (tuple, list)) and len(p) == 2:
                p, opts = p
            elif provider_options and i < len(provider_options):
                opts = provider_options[i]
            if not isinstance(p, str):
                raise InvalidArgument(f"bad provider entry: {p!r}")
            if opts and "device_id" in opts:
                dev = int(opts["device_id"])
    if dev is None:
        dev = int(os.environ.get("MI355VITS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    return dev


def _resolve_devices(spec, default_device: int, library=None) -> List[int]:
    """``SessionOptions.devices`` -> list of device indices (``None``: the single default device)."""
    if spec is None or spec == "" or spec == []:
        return [int(default_device)]
    lib = library or _native.default_library()
    n = lib.device_count()
    if isinstance(spec, str):
        if spec.strip().lower() == "all":
            if n < 1:
                raise RuntimeError("no HIP device available (the MI355X engine has no CPU fallback)")
            return list(range(n))
        try:
            spec = [int(t) for t in spec.replace(";", ",").split(",") if t.strip() != ""]
        except ValueError:
            raise InvalidArgument(f"bad device list {spec!r} (expected e.g. '0,1,2,3' or 'all')") from None
    devs = [int(d) for d in spec]
    if not devs or len(set(devs)) != len(devs) or any(d < 0 or d >= n for d in devs):
        raise InvalidArgument(f"bad device list {devs!r}: {n} device(s) visible")
    return devs


class _LanePool:
    """Engine handles shared by the caller threads, first come first served.  A released handle is handed straight to
    the longest-waiting caller (no barging): with plain lock / queue semantics a worker that has just finished can
    grab the handle again before the notified waiter wakes up, and some callers starve for seconds under load.
    With several devices a caller gets a free lane on the device that has the fewest calls in flight (ties: the device
    used least recently), so concurrent requests spread over the GPUs before they stack up on one."""

    def __init__(self, engines):
        from collections import deque

        self._free = list(engines)
        self._shut = False
        self._waiters = deque()
        self._lock = threading.Lock()
        self.size = len(engines)
        self._busy: Dict[int, int] = {}
        self._last_use: Dict[int, int] = {}
        self._tick = 0
        for e in engines:
            self._busy.setdefault(e.device, 0)
            self._last_use.setdefault(e.device, -1)
        self.calls_per_device: Dict[int, int] = {d: 0 for d in self._busy}

    def _take(self, eng):
        self._tick += 1
        self._busy[eng.device] += 1
        self._last_use[eng.device] = self._tick
        self.calls_per_device[eng.device] += 1
        return eng

    def acquire(self):
        with self._lock:
            if self._shut:
                raise RuntimeError("session is closed")
            if self._free and not self._waiters:
                best = min(range(len(self._free)), key=lambda i: (self._busy[self._free[i].device],
                                                                    self._last_use[self._free[i].device], i))
                return self._take(self._free.pop(best))
            slot = [None]
            ev = threading.Event()
            self._waiters.append((ev, slot))
        try:
            ev.wait()
        except BaseException:
            # interrupted while queued: leave the queue, or give back the lane that was handed over meanwhile
            with self._lock:
                if slot[0] is None:
                    try:
                        self._waiters.remove((ev, slot))
                    except ValueError:
                        pass
                    raise
            self.release(slot[0])
            raise
        if slot[0] is None:  # woken by shutdown(): the session was closed while this caller queued
            raise RuntimeError("session is closed")
        return slot[0]

    def shutdown(self) -> None:
        """After close() has collected every lane: later callers fail at once, callers still queued are woken with an error
        (they would wait forever: no lane is ever released again)."""
        with self._lock:
            self._shut = True
            waiters, self._waiters = list(self._waiters), type(self._waiters)()
        for ev, _slot in waiters:
            ev.set()

    def release(self, eng) -> None:
        with self._lock:
            self._busy[eng.device] -= 1
            if self._waiters:
                ev, slot = self._waiters.popleft()
                slot[0] = self._take(eng)
                ev.set()
            else:
                self._free.append(eng)

    def idle(self) -> int:
        with self._lock:
            return len(self._free)


class _MicroBatcher:
    """Coalesces concurrent B = 1 requests into batched engine calls (one dispatcher thread per session).

    The reference issues one ``run`` per sentence from each of its ``--num-threads`` synthesis workers
    (``mimic3_http/synthesis.py:88-136``); behind an unchanged API this is where batch parallelism comes from."""

    def __init__(self, session: "InferenceSession", window_s: float, max_batch: int):
        import queue
        import weakref

        # weak: the dispatcher thread must not keep a session (and its engines' HBM) alive after its last user is gone
        self._session_ref = weakref.ref(session)
        self._window = window_s
        self._max = max(1, int(max_batch))
        self._q: "queue.Queue" = queue.Queue()
        self.batches = 0
        self.requests = 0
        self._count_lock = threading.Lock()
        # batches handed to a lane and not back yet (the dispatcher holds a new batch back while every lane has one)
        self._inflight = 0
        self._inflight_cv = threading.Condition()
        # with several lanes the dispatcher hands each batch to a worker and goes back to collecting, so that one batch
        # can run its launch-bound front half while the previous one is still in its matrix-core back half
        lanes = len(getattr(session, "_engines", [None]))
        self._lanes = max(1, lanes)
        self._pool = None
        if lanes > 1:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=lanes, thread_name_prefix="mi355vits-lane")
        self._thread = threading.Thread(target=self._loop, name="mi355vits-microbatch", daemon=True)
        self._thread.start()

    def submit(self, ids, lengths, scales, sid, kw):
        from concurrent.futures import Future

        fut: "Future" = Future()
        self._q.put((ids, lengths, scales, sid, kw, fut))
        return fut

    def _loop(self):
        import queue
        import time

        lanes = self._lanes
        while True:
            first = self._q.get()
            if first is None:
                return
            items = [first]
            stop = False

            def take(timeout):  # one more request into `items`; False when there is none (yet) or the batcher is closing
                nonlocal stop
                try:
                    nxt = self._q.get(timeout=timeout) if timeout > 0 else self._q.get_nowait()
                except queue.Empty:
                    return False
                if nxt is None:
                    self._q.put(None)
                    stop = True
                    return False
                items.append(nxt)
                return True

            # (1) the arrival window — skipped for a lone request on an idle session: nothing is in flight and nothing else has
            # arrived, so waiting could only add latency to the very call the reference makes most (one sentence, voice.py:230);
            # a burst then forms behind that first call.  (2) while EVERY lane has a batch, keep collecting instead of queueing
            # small batches behind the lanes: under load a batch is everything that arrived while the chip was busy (round 5's
            # fixed window cut 64 closed-loop clients into batches of 8: profiles/r05_serve_bench.log).
            with self._inflight_cv:
                idle = self._inflight == 0
            while len(items) < self._max and not stop and take(0):  # a backlog (the chip was busy) is a batch already
                pass
            if len(items) == 1 and not idle:
                deadline = time.perf_counter() + self._window
                while len(items) < self._max and not stop:
                    left = deadline - time.perf_counter()
                    if left <= 0 or not take(left):
                        break
            while not stop:
                with self._inflight_cv:
                    if self._inflight < lanes:
                        break
                    if len(items) >= self._max:
                        self._inflight_cv.wait(timeout=0.05)  # a full batch waits for a lane, nothing more to collect
                        continue
                if not take(0.0002):
                    with self._inflight_cv:
                        if self._inflight >= lanes:
                            self._inflight_cv.wait(timeout=0.0002)
            while len(items) < self._max and not stop and take(0):  # whatever arrived meanwhile rides along
                pass
            # only requests with identical scales / output kind can share a call (the ABI takes one `scales`); and only
            # requests of the same phoneme-length class: the text encoder picks its attention / FFN kernels by the padded
            # length (<= 128, 256, 512, beyond), so within a class a row gets the kernels — and the bits — it would get alone
            groups: Dict[Any, list] = {}
            for it in items:
                n = int(it[1][0])
                bucket = 0 if n <= 128 else (1 if n <= 256 else (2 if n <= 512 else 3))
                key = (tuple(np.asarray(it[2], np.float32).tolist()), it[3] is None, tuple(sorted(it[4].items())), bucket)
                groups.setdefault(key, []).append(it)
            for group in groups.values():
                with self._inflight_cv:
                    self._inflight += 1
                if self._pool is not None:
                    self._pool.submit(self._run_counted, group)
                else:
                    self._run_counted(group)

    def _run_counted(self, group):
        released = [False]

        def lane_done():  # the batch's lane is free again: the dispatcher may hand out the next batch
            if not released[0]:
                released[0] = True
                with self._inflight_cv:
                    self._inflight -= 1
                    self._inflight_cv.notify_all()

        try:
            self._run_group(group, lane_done)
        finally:
            lane_done()

    def _run_group(self, group, lane_done=None):
        try:
            tx = max(int(g[0].shape[1]) for g in group)
            B = len(group)
            ids = np.zeros((B, tx), np.int64)
            lens = np.zeros(B, np.int64)
            for b, g in enumerate(group):
                n = int(g[1][0])
                ids[b, :n] = g[0][0, :n]
                lens[b] = n
            sid = None if group[0][3] is None else np.array([int(g[3][0]) for g in group], np.int64)
            kw = dict(group[0][4])
            session = self._session_ref()
            if session is None:
                raise RuntimeError("session closed")
            out = session._engine_run(ids, lens, group[0][2], sid, **kw)
            del session  # (the batch stays "in flight" through the slicing below: handing the next batch out earlier was measured —
            # 19.1k -> 16.9k x real time at 64 clients, the batches shrink from 15.5 to 12.2: profiles/r06_serve_policies.txt)
            with self._count_lock:
                self.batches += 1
                self.requests += B
            for b, g in enumerate(group):
                L = int(out["lengths"][b])
                res = {"lengths": out["lengths"][b:b + 1].copy(), "peaks": out["peaks"][b:b + 1].copy()}
                if "audio" in out:
                    res["audio"] = out["audio"][b:b + 1, :L].copy()
                if "pcm" in out:
                    res["pcm"] = out["pcm"][b:b + 1, :L].copy()
                g[5].set_result(res)
        except BaseException as e:
            if len(group) > 1:
                # a bad request must not poison its batch-mates: fall back to one call per request
                for g in group:
                    if not g[5].done():
                        self._run_group([g], None)
                return
            for g in group:  # every waiter gets its error; nobody hangs
                if not g[5].done():
                    g[5].set_exception(e)

    def close(self):
        import queue

        self._q.put(None)
        on_dispatcher = self._thread is threading.current_thread()
        if not on_dispatcher:
            self._thread.join(timeout=5.0)
        if self._pool is not None:
            self._pool.shutdown(wait=not on_dispatcher)  # batches already handed to a lane finish first
        # requests still queued get an error instead of hanging
        while True:
            try:
                it = self._q.get_nowait()
            except queue.Empty:
                break
            if it is not None and not it[5].done():
                it[5].set_exception(RuntimeError("session closed"))
        if not on_dispatcher and self._thread.is_alive():
            self._q.put(None)  # the drain above may have swallowed the sentinel of a dispatcher that was still busy
            self._thread.join(timeout=5.0)


class InferenceSession:
    """Drop-in for the object stored in ``Mimic3Voice.onnx_model``."""

    _seed_counter = itertools.count(0x5EED)

    def __init__(self, path_or_bytes, sess_options: Optional[SessionOptions] = None, providers=None,
                 provider_options=None, **kwargs):
        self._sess_options = sess_options or SessionOptions()
        self._providers = ["MI355XExecutionProvider"]
        device = _device_from_providers(providers, provider_options, self._sess_options)
        library = kwargs.pop("_library", None)  # tests only: an explicit NativeLibrary
        if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            weights = _model_bytes(bytes(path_or_bytes))
            self._model_path = None
        else:
            weights = resolve_voice_file(path_or_bytes)
            self._model_path = weights if isinstance(weights, str) else os.fspath(path_or_bytes)
        lanes = max(1, int(getattr(self._sess_options, "lanes", 1) or 1))
        devices = _resolve_devices(getattr(self._sess_options, "devices", None), device, library)
        # one weight upload per device; further lanes of a device share it (mi355vits_clone).  Lane-major order:
        # consecutive handles sit on different devices.
        firsts = []
        self._engines = []
        try:
            for d in devices:
                firsts.append(_native.Engine(weights, device=d, library=library))
                self._engines.append(firsts[-1])
            for _ in range(lanes - 1):
                for f in firsts:
                    self._engines.append(f.clone())
            if getattr(self._sess_options, "math", None):
                for e in self._engines:
                    e.set_math(self._sess_options.math)
        except BaseException:
            for e in reversed(self._engines):  # clones before the handles whose weights they share
                e.close()
            self._engines = []
            raise
        self.devices = list(devices)
        self._closed = False
        self._engine = self._engines[0]
        self._free_lanes = _LanePool(self._engines)
        self.config: VitsConfig = self._engine.config
        seed = self._sess_options.seed
        self._seed = int(seed) if seed is not None else next(InferenceSession._seed_counter)
        self._utterances = 0
        self._lock = threading.Lock()
        self.last_lengths: Optional[np.ndarray] = None
        self._batcher: Optional[_MicroBatcher] = None
        if self._sess_options.micro_batch_window_ms and self._sess_options.micro_batch_window_ms > 0:
            self._batcher = _MicroBatcher(self, self._sess_options.micro_batch_window_ms * 1e-3,
                                          self._sess_options.micro_batch_max)

    # ---- onnxruntime surface ----------------------------------------------------------------
    def get_providers(self) -> List[str]:
        return list(self._providers)

    def get_inputs(self) -> List[NodeArg]:
        args = [
            NodeArg("input", "tensor(int64)", ["batch_size", "phonemes"]),
            NodeArg("input_lengths", "tensor(int64)", ["batch_size"]),
            NodeArg("scales", "tensor(float)", [3]),
        ]
        if self.config.is_multispeaker:
            args.append(NodeArg("sid", "tensor(int64)", ["batch_size"]))
        return args

    def get_outputs(self) -> List[NodeArg]:
        return [NodeArg("output", "tensor(float)", ["batch_size", 1, "time"])]

    def run(self, output_names, input_feed: Dict[str, np.ndarray], run_options=None) -> List[np.ndarray]:
        """``run(None, {"input", "input_lengths", "scales"[, "sid"]}) -> [float32 [B, 1, L]]``.

        Rows shorter than the longest one are padding beyond ``last_lengths[b]`` samples (the reference
        only ever calls this with B = 1 and squeezes the result)."""
        if output_names is not None and list(output_names) not in ([], ["output"]):
            raise InvalidArgument(f"unknown output names {output_names!r}; the graph has one output 'output'")
        audio, lengths = self._run(input_feed, want_float=True)["audio"], self.last_lengths
        return [audio[:, None, :]]

    # ---- engine extensions --------------------------------------------------------------------
    def run_pcm16(self, input_feed: Dict[str, np.ndarray], volume: Optional[float] = None, direct: bool = False
                  ) -> Tuple[List[np.ndarray], np.ndarray]:
        """``run`` + ``audio_float_to_int16`` (``utils.py:237-244``) fused on the GPU, per utterance over its
        valid samples.  Returns ([int16 [L_b]] per row, lengths).

        ``volume`` (percent, like ``Mimic3Settings.volume``): additionally applies
        ``audioop.mul(audio_bytes, 2, volume / 100)`` (``tts.py:542-543``) in the same kernel — same bytes as the host
        call, one pass fewer over the audio.  ``direct``: a single-utterance call goes straight to a lane instead
        of through the micro-batcher's queue (a planned request's first sentence: two thread hand-overs fewer; same bits)."""
        kw = {}
        if volume is not None and float(volume) != 100.0:
            if not float(volume) > 0.0:
                raise InvalidArgument("volume must be > 0 (percent)")
            kw["pcm_volume"] = float(volume) / 100.0
        out = self._run(input_feed, _direct=direct, want_float=False, want_pcm16=True, **kw)
        return [out["pcm"][b, : int(out["lengths"][b])] for b in range(out["pcm"].shape[0])], out["lengths"]

    def _run(self, input_feed, _direct: bool = False, **kw) -> Dict[str, np.ndarray]:
        if not isinstance(input_feed, dict):
            raise InvalidArgument("input_feed must be a dict of numpy arrays")
        required = ["input", "input_lengths", "scales"] + (["sid"] if self.config.is_multispeaker else [])
        for name in required:
            if name not in input_feed:
                raise InvalidArgument(f"Required inputs ({name!r}) are missing from input feed ({list(input_feed)}).")
        for name in input_feed:
            if name not in ("input", "input_lengths", "scales", "sid"):
                raise InvalidArgument(f"Invalid input name: {name}")
        ids = np.asarray(input_feed["input"])
        if ids.ndim != 2:
            raise InvalidArgument(f"'input' must have rank 2 [batch, phonemes], got shape {ids.shape}")
        if ids.shape[1] == 0 or ids.shape[0] == 0:
            raise InvalidArgument("'input' must hold at least one phoneme id")
        if not np.issubdtype(ids.dtype, np.integer):
            raise InvalidArgument("'input' must be an int64 tensor")
        sid = input_feed.get("sid") if self.config.is_multispeaker else None
        lengths = np.asarray(input_feed["input_lengths"]).reshape(-1)
        if self._batcher is not None and not _direct and ids.shape[0] == 1 and lengths.shape[0] == 1 and 0 <= int(lengths[0]) <= ids.shape[1]:
            sid1 = None if sid is None else np.asarray(sid).reshape(-1)
            out = self._batcher.submit(np.asarray(ids, np.int64), lengths.astype(np.int64), input_feed["scales"], sid1, kw).result()
        else:
            out = self._engine_run(ids, lengths, input_feed["scales"], sid, **kw)
        self.last_lengths = out["lengths"]
        return out

    def _engine_run(self, ids, lengths, scales, sid, **kw) -> Dict[str, np.ndarray]:
        if self._closed:
            raise RuntimeError("session is closed")
        with self._lock:
            base = self._utterances
            self._utterances += ids.shape[0]
        eng = self._free_lanes.acquire()  # blocks while every lane is busy; first come first served
        try:
            return eng.run(ids, lengths, scales, sid, seed=self._seed, utterance_base=base, **kw)
        except _native.NativeError as e:
            if e.code == -1:
                raise InvalidArgument(str(e)) from None
            raise RuntimeError(str(e)) from None
        finally:
            self._free_lanes.release(eng)

    @property
    def engine(self) -> _native.Engine:
        return self._engine

    def close(self) -> None:
        """Stop the micro-batch dispatcher and release every lane (weights and workspaces in HBM).  Idempotent; also
        runs when the session is garbage-collected."""
        if getattr(self, "_closed", True):
            return
        self._closed = True
        if self._batcher is not None:
            self._batcher.close()
        # wait for every call in flight: a handle is destroyed only once its lane has come back (a worker still inside
        # mi355vits_run must not find its engine freed under it)
        for _ in range(self._free_lanes.size):
            self._free_lanes.acquire()
        self._free_lanes.shutdown()  # a caller that passed the _closed check just before and queued behind us: woken with an error
        for e in reversed(self._engines):  # clones before the handles whose weights they share
            e.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
