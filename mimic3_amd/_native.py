"""ctypes binding of ``libmi355vits.so`` (C ABI: ``include/mi355vits.h``).

The product path loads exactly one file — ``mimic3_amd/csrc/libmi355vits.so`` built by hipcc for
gfx950 — and raises if it is missing or if no HIP device is visible.  There is no CPU fallback.
(``NativeLibrary(path)`` accepts an explicit path only so the test-suite can point the same
binding at its CPU model of the kernels, ``tests/emu/libmi355vits_emu.so``.)
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Dict, Optional, Tuple

import numpy as np

from .config import CVitsConfig, VitsConfig

WANT_FLOAT = 1
WANT_PCM16 = 2
DEVICE_ONLY = 4
DEBUG_TAPS = 8

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "csrc", "libmi355vits.so")
# the product's own objects + the hooks of include/mi355vits_lab.h (kernel unit tests, box probes): test infrastructure and bench.py's
# box probe; the product path (session.py, Engine) never opens it
HOOKS_LIBRARY = os.path.join(_HERE, "csrc", "libmi355vits_hooks.so")

# every symbol include/mi355vits.h declares (the product ABI)
EXPORTED_SYMBOLS = (
    "mi355vits_version", "mi355vits_device_count", "mi355vits_create", "mi355vits_create_from_buffer", "mi355vits_clone", "mi355vits_destroy",
    "mi355vits_device_result", "mi355vits_set_math", "mi355vits_get_math",
    "mi355vits_get_config", "mi355vits_run", "mi355vits_fetch", "mi355vits_free_result",
    "mi355vits_last_error", "mi355vits_profile_enable", "mi355vits_profile_reset",
    "mi355vits_profile_report", "mi355vits_last_run_ms", "mi355vits_get_tap", "mi355vits_get_tap_rows", "mi355vits_list_taps",
)
# every symbol include/mi355vits_lab.h declares: exported by libmi355vits_hooks.so, the lab build and the CPU model — NOT by the product
LAB_SYMBOLS = (
    "mi355vits_test_conv1d", "mi355vits_test_conv_transpose1d", "mi355vits_test_mfma_layout", "mi355vits_bench_conv1d", "mi355vits_probe_device", "mi355vits_probe_weights",
)


class NativeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mi355vits error {code}: {message}")
        self.code = code


class RunArgs(ctypes.Structure):
    _fields_ = [
        ("batch", ctypes.c_int32),
        ("tx_max", ctypes.c_int32),
        ("ids", ctypes.POINTER(ctypes.c_int64)),
        ("lengths", ctypes.POINTER(ctypes.c_int64)),
        ("scales", ctypes.POINTER(ctypes.c_float)),
        ("sid", ctypes.POINTER(ctypes.c_int64)),
        ("seed", ctypes.c_uint64),
        ("utterance_base", ctypes.c_uint64),
        ("noise_w", ctypes.POINTER(ctypes.c_float)),
        ("noise_z", ctypes.POINTER(ctypes.c_float)),
        ("noise_z_frames", ctypes.c_int32),
        ("forced_durations", ctypes.POINTER(ctypes.c_int32)),
        ("flags", ctypes.c_uint32),
        ("pcm_volume", ctypes.c_double),
    ]


class Result(ctypes.Structure):
    _fields_ = [
        ("batch", ctypes.c_int32),
        ("l_max", ctypes.c_int64),
        ("ty_max", ctypes.c_int64),
        ("audio", ctypes.POINTER(ctypes.c_float)),
        ("pcm", ctypes.POINTER(ctypes.c_int16)),
        ("lengths", ctypes.POINTER(ctypes.c_int64)),
        ("peaks", ctypes.POINTER(ctypes.c_float)),
        ("owner_", ctypes.c_void_p),
    ]


class ConvTest(ctypes.Structure):
    _fields_ = [
        ("impl", ctypes.c_int32), ("B", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("T", ctypes.c_int32), ("K", ctypes.c_int32), ("dilation", ctypes.c_int32),
        ("x", ctypes.POINTER(ctypes.c_float)), ("w", ctypes.POINTER(ctypes.c_float)),
        ("bias", ctypes.POINTER(ctypes.c_float)), ("res", ctypes.POINTER(ctypes.c_float)),
        ("in_len", ctypes.POINTER(ctypes.c_int32)), ("out_len", ctypes.POINTER(ctypes.c_int32)),
        ("in_slope", ctypes.c_float), ("relu", ctypes.c_int32), ("out_scale", ctypes.c_float),
        ("res_sub", ctypes.c_int32), ("y", ctypes.POINTER(ctypes.c_float)), ("accumulate", ctypes.c_int32),
    ]


def _fptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class NativeLibrary:
    """One loaded copy of the C ABI."""

    def __init__(self, path: Optional[str] = None):
        self.path = path or DEFAULT_LIBRARY
        if not os.path.exists(self.path):
            raise RuntimeError(
                f"native library not found: {self.path}. Build it with `python -m mimic3_amd.build hip` "
                "(hipcc --offload-arch=gfx950). The MI355X engine has no CPU fallback."
            )
        self.lib = ctypes.CDLL(self.path)
        L = self.lib
        for sym in EXPORTED_SYMBOLS:
            if not hasattr(L, sym):
                raise RuntimeError(f"{self.path} does not export {sym}")
        H = ctypes.c_void_p
        L.mi355vits_version.restype = ctypes.c_char_p
        L.mi355vits_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(H)]
        L.mi355vits_create_from_buffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(H)]
        L.mi355vits_clone.argtypes = [H, ctypes.POINTER(H)]
        L.mi355vits_device_result.argtypes = [H, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                              ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32),
                                              ctypes.POINTER(ctypes.c_void_p)]
        L.mi355vits_set_math.argtypes = [H, ctypes.c_int]
        L.mi355vits_get_math.argtypes = [H]
        L.mi355vits_destroy.argtypes = [H]
        L.mi355vits_destroy.restype = None
        L.mi355vits_get_config.argtypes = [H, ctypes.POINTER(CVitsConfig)]
        L.mi355vits_run.argtypes = [H, ctypes.POINTER(RunArgs), ctypes.POINTER(Result)]
        L.mi355vits_fetch.argtypes = [H, ctypes.c_uint32, ctypes.POINTER(Result)]
        L.mi355vits_free_result.argtypes = [ctypes.POINTER(Result)]
        L.mi355vits_free_result.restype = None
        L.mi355vits_last_error.argtypes = [H]
        L.mi355vits_last_error.restype = ctypes.c_char_p
        L.mi355vits_profile_enable.argtypes = [H, ctypes.c_int]
        L.mi355vits_profile_reset.argtypes = [H]
        L.mi355vits_profile_report.argtypes = [H, ctypes.c_char_p, ctypes.c_size_t]
        L.mi355vits_profile_report.restype = ctypes.c_long
        L.mi355vits_last_run_ms.argtypes = [H]
        L.mi355vits_last_run_ms.restype = ctypes.c_float
        L.mi355vits_get_tap.argtypes = [H, ctypes.c_char_p, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_int64)]
        L.mi355vits_get_tap.restype = ctypes.c_long
        L.mi355vits_get_tap_rows.argtypes = [H, ctypes.c_char_p, ctypes.c_long, ctypes.c_long, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t,
                                            ctypes.POINTER(ctypes.c_int64)]
        L.mi355vits_get_tap_rows.restype = ctypes.c_long
        L.mi355vits_list_taps.argtypes = [H, ctypes.c_char_p, ctypes.c_size_t]
        L.mi355vits_list_taps.restype = ctypes.c_long
        # the hooks of include/mi355vits_lab.h: all of them or none (the product library has none)
        present = [hasattr(L, sym) for sym in LAB_SYMBOLS]
        if any(present) and not all(present):
            raise RuntimeError(f"{self.path} exports only part of include/mi355vits_lab.h")
        self.has_hooks = all(present)
        if self.has_hooks:
            L.mi355vits_test_conv1d.argtypes = [ctypes.c_int, ctypes.POINTER(ConvTest)]
            L.mi355vits_test_conv_transpose1d.argtypes = [
                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                ctypes.POINTER(ctypes.c_float), ctypes.c_float, ctypes.POINTER(ctypes.c_float)]
            L.mi355vits_test_mfma_layout.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
            L.mi355vits_bench_conv1d.argtypes = [ctypes.c_int] * 9 + [ctypes.POINTER(ctypes.c_float)]
            L.mi355vits_probe_device.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
            L.mi355vits_probe_weights.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]

    def _need_hooks(self):
        if not self.has_hooks:
            raise RuntimeError(f"{self.path} is the product library: the hooks of include/mi355vits_lab.h live in "
                               "libmi355vits_hooks.so (mimic3_amd._native.hooks_library()), the lab build and the CPU model")

    def version(self) -> str:
        return self.lib.mi355vits_version().decode()

    def device_count(self) -> int:
        return int(self.lib.mi355vits_device_count())

    def create_error(self) -> str:
        return (self.lib.mi355vits_last_error(None) or b"").decode("utf-8", "replace")

    # ---- kernel unit-test hooks -------------------------------------------------------------
    def test_conv1d(self, x, w, bias=None, res=None, dilation=1, impl=1, in_len=None, out_len=None, in_slope=1.0,
                    relu=False, out_scale=1.0, res_sub=False, accumulate_into=None, device=0) -> np.ndarray:
        self._need_hooks()
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        B, Cin, T = x.shape
        Cout, _, K = w.shape
        y = np.zeros((B, Cout, T), np.float32) if accumulate_into is None else np.ascontiguousarray(accumulate_into, np.float32).copy()
        keep = [x, w, y]
        t = ConvTest()
        t.impl, t.B, t.Cin, t.Cout, t.T, t.K, t.dilation = impl, B, Cin, Cout, T, K, dilation
        t.x, t.w, t.y = _fptr(x), _fptr(w), _fptr(y)
        for name, arr in (("bias", bias), ("res", res)):
            if arr is not None:
                a = np.ascontiguousarray(arr, np.float32)
                keep.append(a)
                setattr(t, name, _fptr(a))
        for name, arr in (("in_len", in_len), ("out_len", out_len)):
            if arr is not None:
                a = np.ascontiguousarray(arr, np.int32)
                keep.append(a)
                setattr(t, name, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        t.in_slope, t.relu, t.out_scale, t.res_sub = in_slope, int(relu), out_scale, int(res_sub)
        t.accumulate = int(accumulate_into is not None)
        rc = self.lib.mi355vits_test_conv1d(device, ctypes.byref(t))
        if rc != 0:
            raise NativeError(rc, self.create_error())
        return y

    def test_conv_transpose1d(self, x, w, bias, stride, in_slope=1.0, device=0, impl=0) -> np.ndarray:
        self._need_hooks()
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        B, Cin, Tin = x.shape
        _, Cout, K = w.shape
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        y = np.zeros((B, Cout, Tin * stride), np.float32)
        rc = self.lib.mi355vits_test_conv_transpose1d(device, impl, B, Cin, Cout, Tin, K, stride, _fptr(x), _fptr(w),
                                                      _fptr(b), in_slope, _fptr(y))
        if rc != 0:
            raise NativeError(rc, self.create_error())
        return y

    def bench_conv1d(self, B, Cin, Cout, T, K, dilation=1, epi=0, reps=20, device=0) -> float:
        self._need_hooks()
        ms = ctypes.c_float(-1.0)
        rc = self.lib.mi355vits_bench_conv1d(device, B, Cin, Cout, T, K, dilation, epi, reps, ctypes.byref(ms))
        if rc != 0:
            raise NativeError(rc, self.create_error())
        return float(ms.value)

    def probe_device(self, device=0) -> dict:
        """The box probe: what this lease's chip gives the kernels' access patterns (include/mi355vits_lab.h)."""
        self._need_hooks()
        out = (ctypes.c_double * 8)()
        rc = self.lib.mi355vits_probe_device(device, out)
        if rc != 0:
            raise RuntimeError(f"mi355vits_probe_device failed: rc={rc}")
        return {"l2_stream_GBps": round(out[0], 1), "l2_hit_latency_ns": round(out[1], 1), "hbm_copy_GBps": round(out[2], 1), "cus": int(out[3]),
                "l2_stream_beside_copy_GBps": round(out[4], 1), "table_24MB_stream_GBps": round(out[5], 1),
                "latency_32MB_ns": round(out[6], 1), "latency_1GiB_ns": round(out[7], 1)}

    def test_mfma_layout(self, device=0) -> float:
        self._need_hooks()
        err = ctypes.c_float(-1.0)
        rc = self.lib.mi355vits_test_mfma_layout(device, ctypes.byref(err))
        if rc != 0:
            raise NativeError(rc, self.create_error() + f" (max err {err.value})")
        return float(err.value)


_default_lock = threading.Lock()
_default: Optional[NativeLibrary] = None


def default_library() -> NativeLibrary:
    """The product library (gfx950).  Raises when it has not been built."""
    global _default
    with _default_lock:
        if _default is None:
            _default = NativeLibrary(DEFAULT_LIBRARY)
        return _default


_hooks: Optional[NativeLibrary] = None


def hooks_library() -> NativeLibrary:
    """libmi355vits_hooks.so: the product's own object files + the hooks of include/mi355vits_lab.h (kernel unit tests, conv
    micro-benchmark, box probes).  Test infrastructure and bench.py's box probe; nothing on the product path opens it."""
    global _hooks
    with _default_lock:
        if _hooks is None:
            _hooks = NativeLibrary(HOOKS_LIBRARY)
            if not _hooks.has_hooks:
                raise RuntimeError(f"{HOOKS_LIBRARY} does not export include/mi355vits_lab.h")
        return _hooks


class _ResultHolder:
    """Keeps one ``mi355vits_result`` alive for the numpy views made of it; releases it when they are gone."""

    def __init__(self, native: "NativeLibrary", r: Result):
        self._native = native
        self._r = Result()
        ctypes.memmove(ctypes.byref(self._r), ctypes.byref(r), ctypes.sizeof(Result))

    def view(self, ptr, ctype, dtype, B: int, L: int) -> np.ndarray:
        n = int(B) * int(L)
        buf = (ctype * n).from_address(ctypes.addressof(ptr.contents))
        buf._holder = self  # the array's base chain (memoryview -> ctypes array) keeps the holder alive
        return np.frombuffer(buf, dtype=dtype, count=n).reshape(B, L)

    def __del__(self):
        try:
            self._native.lib.mi355vits_free_result(ctypes.byref(self._r))
        except Exception:
            pass


class Engine:
    """A voice loaded on one GPU (wraps ``mi355vits_handle``)."""

    def __init__(self, weights, device: int = 0, library: Optional[NativeLibrary] = None):
        """``weights``: path to an ``.m355`` container, its bytes, or another ``Engine`` — then this is a further lane
        on that engine's device sharing its weight replica (``mi355vits_clone``; ``device`` is ignored)."""
        self.native = weights.native if isinstance(weights, Engine) else (library or default_library())
        self._h = ctypes.c_void_p()
        L = self.native.lib
        if isinstance(weights, Engine):
            rc = L.mi355vits_clone(weights._h, ctypes.byref(self._h))
            if rc != 0:
                self._h = ctypes.c_void_p()
                raise NativeError(rc, (L.mi355vits_last_error(weights._h) or b"").decode("utf-8", "replace"))
            device = weights.device
        elif isinstance(weights, (bytes, bytearray, memoryview)):
            buf = bytes(weights)
            rc = L.mi355vits_create_from_buffer(buf, len(buf), device, ctypes.byref(self._h))
        else:
            rc = L.mi355vits_create(os.fsencode(str(weights)), device, ctypes.byref(self._h))
        if rc != 0:
            self._h = ctypes.c_void_p()
            raise NativeError(rc, self.native.create_error())
        c = CVitsConfig()
        self._check(L.mi355vits_get_config(self._h, ctypes.byref(c)))
        self.config = VitsConfig.from_c(c)
        self.device = device

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise NativeError(rc, (self.native.lib.mi355vits_last_error(self._h) or b"").decode("utf-8", "replace"))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.native.lib.mi355vits_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one synthesis call ---------------------------------------------------------------------
    def run(self, ids, lengths, scales, sid=None, *, seed: int = 0, utterance_base: int = 0, noise_w=None,
            noise_z=None, forced_durations=None, want_float: bool = True, want_pcm16: bool = False,
            device_only: bool = False, debug_taps: bool = False, pcm_volume: float = 1.0) -> Dict[str, np.ndarray]:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        if ids.ndim != 2:
            raise ValueError("'input' must have shape [batch, phonemes]")
        B, Tx = ids.shape
        lengths = np.ascontiguousarray(lengths, dtype=np.int64).reshape(-1)
        if lengths.shape[0] != B:
            raise ValueError("'input_lengths' must have shape [batch]")
        scales = np.ascontiguousarray(scales, dtype=np.float32).reshape(-1)
        if scales.shape[0] != 3:
            raise ValueError("'scales' must hold [noise_scale, length_scale, noise_w]")
        a = RunArgs()
        keep = [ids, lengths, scales]
        a.batch, a.tx_max = B, Tx
        a.ids = ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        a.lengths = lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        a.scales = _fptr(scales)
        if sid is not None:
            sid = np.ascontiguousarray(sid, dtype=np.int64).reshape(-1)
            if sid.shape[0] != B:
                raise ValueError("'sid' must have shape [batch]")
            keep.append(sid)
            a.sid = sid.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        a.seed = seed & 0xFFFFFFFFFFFFFFFF
        a.utterance_base = utterance_base
        if noise_w is not None:
            noise_w = np.ascontiguousarray(noise_w, dtype=np.float32)
            if noise_w.shape != (B, 2, Tx):
                raise ValueError("noise_w must have shape [batch, 2, phonemes]")
            keep.append(noise_w)
            a.noise_w = _fptr(noise_w)
        if noise_z is not None:
            noise_z = np.ascontiguousarray(noise_z, dtype=np.float32)
            if noise_z.ndim != 3 or noise_z.shape[:2] != (B, self.config.inter_channels):
                raise ValueError("noise_z must have shape [batch, inter_channels, frames]")
            keep.append(noise_z)
            a.noise_z = _fptr(noise_z)
            a.noise_z_frames = noise_z.shape[2]
        if forced_durations is not None:
            forced_durations = np.ascontiguousarray(forced_durations, dtype=np.int32)
            if forced_durations.shape != (B, Tx):
                raise ValueError("forced_durations must have shape [batch, phonemes]")
            keep.append(forced_durations)
            a.forced_durations = forced_durations.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        a.flags = (WANT_FLOAT if want_float else 0) | (WANT_PCM16 if want_pcm16 else 0) | \
                  (DEVICE_ONLY if device_only else 0) | (DEBUG_TAPS if debug_taps else 0)
        a.pcm_volume = float(pcm_volume)
        r = Result()
        self._check(self.native.lib.mi355vits_run(self._h, ctypes.byref(a), ctypes.byref(r)))
        del keep
        return self._take(r)

    MATH_MODES = {"f32": 0, "bf16x3": 1, "bf16w": 2, "f16x2": 3}

    def set_math(self, mode) -> None:
        """``"f32"`` (f32 MFMA), ``"bf16x3"`` (f32 operands split 3 x bf16, six bf16-MFMA products, f32 accumulate; the
        default), ``"bf16w"`` (bf16-rounded weights x exact activations: reduced precision, BASELINE configs[4]) or ``"f16x2"``
        (experimental, fixed-scale: every kernel of the bf16x3 split — fused MRF stages, WaveNet layers, staged convs, upsamplers —
        with operands as two fp16 terms = 22 significant bits, three products; activations beyond |x| = 4094 clip, a stage with a
        weight |w| >= 7.99 runs as bf16x3; the text encoder / duration predictor as in bf16x3.  See include/mi355vits.h)."""
        self._check(self.native.lib.mi355vits_set_math(self._h, self.MATH_MODES.get(mode, mode)))

    @property
    def math(self) -> str:
        m = int(self.native.lib.mi355vits_get_math(self._h))
        return {v: k for k, v in self.MATH_MODES.items()}.get(m, str(m))

    def clone(self) -> "Engine":
        """Another lane on this engine's device: own stream + workspace, shared weights."""
        return Engine(self)

    def device_result(self) -> Dict[str, object]:
        """Device pointers of the last run's results (``mi355vits_device_result``): ``{"pcm": ptr, "audio": ptr,
        "row_stride": L, "batch": B, "lengths": ptr, "device": d}`` — valid until the next run on this handle."""
        pcm, audio, lens = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        rs, b = ctypes.c_int64(), ctypes.c_int32()
        self._check(self.native.lib.mi355vits_device_result(self._h, ctypes.byref(pcm), ctypes.byref(audio), ctypes.byref(rs),
                                                           ctypes.byref(b), ctypes.byref(lens)))
        return {"pcm": pcm.value, "audio": audio.value, "row_stride": int(rs.value), "batch": int(b.value),
                "lengths": lens.value, "device": self.device}

    def fetch(self, want_float: bool = True, want_pcm16: bool = False) -> Dict[str, np.ndarray]:
        r = Result()
        flags = (WANT_FLOAT if want_float else 0) | (WANT_PCM16 if want_pcm16 else 0)
        self._check(self.native.lib.mi355vits_fetch(self._h, flags, ctypes.byref(r)))
        return self._take(r)

    def _take(self, r: Result) -> Dict[str, np.ndarray]:
        """Result struct -> numpy.  The waveform arrays are *views of the callee's pinned buffers* that own them: the
        buffers go back to the library (``mi355vits_free_result`` -> pinned pool) when the last array referring to
        them is garbage-collected.  No host-side copy of the audio, no aliasing between calls (a buffer is handed out
        exclusively until released) — "a new ndarray owned by Python", as ``onnx_model.run`` returns."""
        B, L = r.batch, r.l_max
        try:
            out: Dict[str, np.ndarray] = {
                "lengths": np.ctypeslib.as_array(r.lengths, shape=(B,)).copy(),
                "peaks": np.ctypeslib.as_array(r.peaks, shape=(B,)).copy(),
                "l_max": np.int64(L), "ty_max": np.int64(r.ty_max),
            }
        except BaseException:
            self.native.lib.mi355vits_free_result(ctypes.byref(r))
            raise
        if not r.audio and not r.pcm:
            self.native.lib.mi355vits_free_result(ctypes.byref(r))
            return out
        holder = _ResultHolder(self.native, r)
        if r.audio:
            out["audio"] = holder.view(r.audio, ctypes.c_float, np.float32, B, L)
        if r.pcm:
            out["pcm"] = holder.view(r.pcm, ctypes.c_int16, np.int16, B, L)
        return out

    # ---- profiling / debugging ------------------------------------------------------------------
    def last_run_ms(self) -> float:
        return float(self.native.lib.mi355vits_last_run_ms(self._h))

    def probe_weights(self) -> dict:
        """L2 stream over this replica's own weight arena (include/mi355vits_lab.h mi355vits_probe_weights; the handle must come
        from a library that carries the hooks): GB/s min / median / max over 2.6 MB windows, eight loads in flight per lane and one."""
        self.native._need_hooks()
        out = (ctypes.c_double * 8)()
        self._check(self.native.lib.mi355vits_probe_weights(self._h, out))
        return {"arena_stream8_GBps": [round(out[0]), round(out[1]), round(out[2])], "arena_stream1_GBps": [round(out[3]), round(out[4]), round(out[5])],
                "windows": int(out[6]), "arena_addr_low36": hex(int(out[7]))}

    def profile_enable(self, on: bool = True) -> None:
        self._check(self.native.lib.mi355vits_profile_enable(self._h, int(on)))

    def profile_reset(self) -> None:
        self._check(self.native.lib.mi355vits_profile_reset(self._h))

    def profile_report(self) -> Dict[str, Dict[str, float]]:
        buf = ctypes.create_string_buffer(1 << 16)
        n = self.native.lib.mi355vits_profile_report(self._h, buf, len(buf))
        if n < 0:
            self._check(int(n))
        rep = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms, flops, nbytes = line.split()
            rep[name] = {"calls": int(calls), "ms": float(ms), "flops": float(flops), "bytes": float(nbytes)}
        return rep

    def taps(self):
        buf = ctypes.create_string_buffer(1 << 14)
        self.native.lib.mi355vits_list_taps(self._h, buf, len(buf))
        return [t for t in buf.value.decode().splitlines() if t]

    def tap(self, name: str, row0: int = 0, nrows: int = -1) -> np.ndarray:
        """The named intermediate of the last ``debug_taps`` run as [B, C, T]; ``row0`` / ``nrows``: those batch rows only."""
        dims = (ctypes.c_int64 * 4)()
        L = self.native.lib
        if nrows < 0:
            get = lambda out, cap: L.mi355vits_get_tap(self._h, name.encode(), out, cap, dims)  # noqa: E731
        else:
            get = lambda out, cap: L.mi355vits_get_tap_rows(self._h, name.encode(), row0, nrows, out, cap, dims)  # noqa: E731
        n = get(None, 0)
        if n < 0:
            self._check(int(n))
        out = np.empty(int(n), np.float32)
        n2 = get(_fptr(out), out.size)
        if n2 < 0:
            self._check(int(n2))
        shape = [int(d) for d in dims][:3]  # taps are [B, C, T]
        return out.reshape(shape)
