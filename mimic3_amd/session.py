"""Host-side mirror of the one interface Mimic 3 uses on its hot path.

``Mimic3Voice`` keeps an ``onnxruntime.InferenceSession`` in ``self.onnx_model`` and calls exactly
``self.onnx_model.run(None, inputs)[0].squeeze()`` (``mimic3_tts/voice.py:230``); the session is built in
``Mimic3Voice._load_model`` (``voice.py:378-407``) from ``onnxruntime.SessionOptions()`` with
``graph_optimization_level`` / ``use_deterministic_compute`` assigned and ``providers=None`` or
``["CUDAExecutionProvider"]`` (``tts.py:590-593``).  The classes below keep those names, argument
meanings and error behaviour (Python exceptions, never partial audio) and forward to the MI355X engine
through the C ABI (``include/mi355vits.h``).  There is no CPU execution path here.

Voice files: the reference passes ``<voice_dir>/generator.onnx`` (``voice.py:273``).  The engine reads the
voice from ``generator.m355`` beside it (weights + hyper-parameters, written by
``mimic3_amd.weights.save``); a path that already ends in ``.m355`` is used as is.
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


class NodeArg:
    def __init__(self, name: str, type_: str, shape: Sequence[Any]):
        self.name = name
        self.type = type_
        self.shape = list(shape)

    def __repr__(self):
        return f"NodeArg(name='{self.name}', type='{self.type}', shape={self.shape})"


class InvalidArgument(ValueError):
    """Bad feed (what onnxruntime reports as ``InvalidArgument``)."""


def resolve_voice_file(path: Union[str, os.PathLike]) -> str:
    p = os.fspath(path)
    if p.endswith(".m355"):
        cand = p
    else:
        cand = os.path.splitext(p)[0] + ".m355"
    if not os.path.isfile(cand):
        raise FileNotFoundError(
            f"{cand} not found. The MI355X engine loads voices from an .m355 container placed beside "
            f"generator.onnx (see INTEGRATION.md); converting ONNX initialisers is not available in this build."
        )
    return cand


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
            weights = bytes(path_or_bytes)
            self._model_path = None
        else:
            self._model_path = resolve_voice_file(path_or_bytes)
            weights = self._model_path
        self._engine = _native.Engine(weights, device=device, library=library)
        self.config: VitsConfig = self._engine.config
        seed = self._sess_options.seed
        self._seed = int(seed) if seed is not None else next(InferenceSession._seed_counter)
        self._utterances = 0
        self._lock = threading.Lock()
        self.last_lengths: Optional[np.ndarray] = None

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
    def run_pcm16(self, input_feed: Dict[str, np.ndarray]) -> Tuple[List[np.ndarray], np.ndarray]:
        """``run`` + ``audio_float_to_int16`` (``utils.py:237-244``) fused on the GPU, per utterance over its
        valid samples.  Returns ([int16 [L_b]] per row, lengths)."""
        out = self._run(input_feed, want_float=False, want_pcm16=True)
        return [out["pcm"][b, : int(out["lengths"][b])] for b in range(out["pcm"].shape[0])], out["lengths"]

    def _run(self, input_feed, **kw) -> Dict[str, np.ndarray]:
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
        with self._lock:
            base = self._utterances
            self._utterances += ids.shape[0]
        try:
            out = self._engine.run(ids, input_feed["input_lengths"], input_feed["scales"], sid, seed=self._seed,
                                   utterance_base=base, **kw)
        except _native.NativeError as e:
            if e.code == -1:
                raise InvalidArgument(str(e)) from None
            raise RuntimeError(str(e)) from None
        self.last_lengths = out["lengths"]
        return out

    @property
    def engine(self) -> _native.Engine:
        return self._engine
