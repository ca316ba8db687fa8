"""Hyper-parameter surface of a Mimic 3 VITS voice.

Mirrors the fields of the reference's ``ModelConfig`` / ``AudioConfig`` /
``InferenceConfig`` (``mimic3_tts/config.py:112-143``, ``:30-60``, ``:256-271``)
that shape the inference graph executed behind ``voice.py:230``.  The values a
voice ships in its ``config.json`` are read with :meth:`VitsConfig.from_json`.

The reference dataclasses depend on ``dataclasses_json`` / ``gruut_ipa`` /
``phonemes2ids`` (absent here) and carry training-only fields; this mirror keeps
only what the engine and the oracle need and adds the few architecture
constants that upstream VITS hard-codes (relative window, flow depth, spline
bins) so that both sides are driven by one description.
"""
from __future__ import annotations

import ctypes
import json
from dataclasses import dataclass, field, asdict
from typing import Tuple

MAX_STAGES = 8  # upsample stages / resblock kernels / dilations per resblock


@dataclass
class VitsConfig:
    # --- ModelConfig (mimic3_tts/config.py:112-143) ---
    num_symbols: int = 50
    n_speakers: int = 1
    inter_channels: int = 192
    hidden_channels: int = 192
    filter_channels: int = 768
    n_heads: int = 2
    n_layers: int = 6
    kernel_size: int = 3
    resblock: str = "2"
    resblock_kernel_sizes: Tuple[int, ...] = (3, 5, 7)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 2), (2, 6), (3, 12))
    upsample_rates: Tuple[int, ...] = (8, 8, 4)
    upsample_initial_channel: int = 256
    upsample_kernel_sizes: Tuple[int, ...] = (16, 16, 8)
    gin_channels: int = 0
    use_sdp: bool = True
    # --- constants hard-coded by upstream VITS ---
    window_size: int = 4            # relative-position window (A.4)
    flow_n_flows: int = 4           # residual coupling layers (A.9)
    flow_wn_layers: int = 4
    flow_wn_kernel: int = 5
    flow_wn_dilation_rate: int = 1
    dp_kernel_size: int = 3         # SDP DDSConv kernel (A.6)
    dp_dds_layers: int = 3
    dp_n_flows: int = 4             # ConvFlows; inference uses n-1 (A.7)
    dp_num_bins: int = 10
    dp_tail_bound: float = 5.0
    # --- AudioConfig (mimic3_tts/config.py:30-60) ---
    sample_rate: int = 22050
    hop_length: int = 256
    # --- InferenceConfig (mimic3_tts/config.py:256-271) ---
    length_scale: float = 1.0
    noise_scale: float = 0.667
    noise_w: float = 0.8

    # ------------------------------------------------------------------ helpers
    @property
    def is_multispeaker(self) -> bool:
        """`ModelConfig.is_multispeaker` (config.py:141-143)."""
        return self.n_speakers > 1

    @property
    def half_channels(self) -> int:
        return self.inter_channels // 2

    @property
    def upsample_factor(self) -> int:
        f = 1
        for r in self.upsample_rates:
            f *= r
        return f

    def validate(self) -> None:
        if self.resblock not in ("1", "2"):
            raise ValueError(f"resblock must be '1' or '2', got {self.resblock!r}")
        if len(self.upsample_rates) != len(self.upsample_kernel_sizes):
            raise ValueError("upsample_rates / upsample_kernel_sizes length mismatch")
        if len(self.resblock_kernel_sizes) != len(self.resblock_dilation_sizes):
            raise ValueError("resblock_kernel_sizes / resblock_dilation_sizes length mismatch")
        if max(len(self.upsample_rates), len(self.resblock_kernel_sizes)) > MAX_STAGES:
            raise ValueError("too many decoder stages")
        for d in self.resblock_dilation_sizes:
            if len(d) > MAX_STAGES:
                raise ValueError("too many dilations in a resblock")
        if self.hidden_channels % self.n_heads:
            raise ValueError("hidden_channels must be divisible by n_heads")
        if self.inter_channels % 2:
            raise ValueError("inter_channels must be even")
        for r, k in zip(self.upsample_rates, self.upsample_kernel_sizes):
            if (k - r) % 2:
                raise ValueError("upsample kernel - rate must be even")
        if self.is_multispeaker and self.gin_channels <= 0:
            raise ValueError("multi-speaker voice needs gin_channels > 0")
        if not self.use_sdp:
            raise ValueError("only the stochastic duration predictor is supported (use_sdp=True)")

    # the two voices named by BASELINE.json, pinned in SURVEY.md §8a-0
    @staticmethod
    def apope_low(num_symbols: int = 50) -> "VitsConfig":
        return VitsConfig(num_symbols=num_symbols)

    @staticmethod
    def vctk_low(num_symbols: int = 50) -> "VitsConfig":
        return VitsConfig(num_symbols=num_symbols, n_speakers=109, gin_channels=512)

    @staticmethod
    def tiny(num_symbols: int = 20, n_speakers: int = 1, resblock: str = "2") -> "VitsConfig":
        """A shrunken graph of the same topology, for fast CPU tests."""
        return VitsConfig(
            num_symbols=num_symbols,
            n_speakers=n_speakers,
            inter_channels=32,
            hidden_channels=32,
            filter_channels=64,
            n_heads=2,
            n_layers=2,
            resblock=resblock,
            resblock_kernel_sizes=(3, 5) if resblock == "2" else (3, 5),
            resblock_dilation_sizes=((1, 2), (2, 6)) if resblock == "2" else ((1, 3), (1, 3)),
            upsample_rates=(4, 2),
            upsample_initial_channel=32,
            upsample_kernel_sizes=(8, 4),
            gin_channels=16 if n_speakers > 1 else 0,
            flow_n_flows=2,
            flow_wn_layers=2,
            hop_length=8,
        )

    @staticmethod
    def tiny_wide(n_speakers: int = 1, initial_channel: int = 128) -> "VitsConfig":
        """Tiny encoder/flow with the *real* decoder widths of the last two stages (64 -> 32 channels, ResBlock2
        k = 3/5/7, dilations (1,2)/(2,6)/(3,12)) so that the fused multi-receptive-field kernel is exercised."""
        c = VitsConfig.tiny(n_speakers=n_speakers)
        c.upsample_initial_channel = initial_channel  # 256: stages of 128 and 64 channels, like the first two real ones
        c.resblock_kernel_sizes = (3, 5, 7)
        c.resblock_dilation_sizes = ((1, 2), (2, 6), (3, 12))
        return c

    @staticmethod
    def tiny_h192(n_speakers: int = 1) -> "VitsConfig":
        """Tiny decoder / depth with the real hidden width (192): exercises the 6-wave fused WaveNet-layer kernel
        and the head-dimension-96 attention on the CPU model."""
        c = VitsConfig.tiny(n_speakers=n_speakers)
        c.hidden_channels = 192
        c.n_layers = 1
        return c

    # ------------------------------------------------------------------ (de)serialisation
    @staticmethod
    def from_json(text_or_dict) -> "VitsConfig":
        """Read a voice ``config.json`` (the layout written by
        ``TrainingConfig.save``, config.py:320-322: top-level ``model``, ``audio``,
        ``inference`` objects). Unknown keys are ignored."""
        d = json.loads(text_or_dict) if isinstance(text_or_dict, (str, bytes)) else dict(text_or_dict)
        model = d.get("model", d)
        audio = d.get("audio", {})
        infer = d.get("inference", {})
        cfg = VitsConfig()
        # A key that config.json leaves out takes the REFERENCE's default (ModelConfig, mimic3_tts/config.py:112-139: the
        # "high quality" graph — ResBlock1, kernels 3/7/11, upsampling 8-8-2-2 from 512 channels), not this engine's
        # "low" defaults: the reference would build exactly that graph from the same file.
        cfg.resblock = "1"
        cfg.resblock_kernel_sizes = (3, 7, 11)
        cfg.resblock_dilation_sizes = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
        cfg.upsample_rates = (8, 8, 2, 2)
        cfg.upsample_initial_channel = 512
        cfg.upsample_kernel_sizes = (16, 16, 4, 4)
        cfg.declared_model_keys = frozenset(k for k in model if isinstance(k, str))  # what the file actually says
        for k in (
            "num_symbols", "n_speakers", "inter_channels", "hidden_channels", "filter_channels",
            "n_heads", "n_layers", "kernel_size", "resblock", "upsample_initial_channel",
            "gin_channels", "use_sdp",
        ):
            if k in model:
                setattr(cfg, k, model[k])
        cfg.resblock = str(cfg.resblock)
        for k in ("resblock_kernel_sizes", "upsample_rates", "upsample_kernel_sizes"):
            if k in model:
                setattr(cfg, k, tuple(int(v) for v in model[k]))
        if "resblock_dilation_sizes" in model:
            cfg.resblock_dilation_sizes = tuple(tuple(int(v) for v in row) for row in model["resblock_dilation_sizes"])
        for k in ("sample_rate", "hop_length"):
            if k in audio:
                setattr(cfg, k, int(audio[k]))
        for k in ("length_scale", "noise_scale", "noise_w"):
            if k in infer and infer[k] is not None:
                setattr(cfg, k, float(infer[k]))
        for k, v in d.get("vits_constants", {}).items():  # written by to_json(); absent in a trainer config.json
            if hasattr(cfg, k):
                setattr(cfg, k, type(getattr(cfg, k))(v))
        cfg.validate()
        return cfg

    def to_json(self) -> str:
        d = asdict(self)
        model_keys = {
            "num_symbols", "n_speakers", "inter_channels", "hidden_channels", "filter_channels",
            "n_heads", "n_layers", "kernel_size", "resblock", "resblock_kernel_sizes",
            "resblock_dilation_sizes", "upsample_rates", "upsample_initial_channel",
            "upsample_kernel_sizes", "gin_channels", "use_sdp",
        }
        out = {
            "model": {k: d[k] for k in model_keys},
            "audio": {"sample_rate": self.sample_rate, "hop_length": self.hop_length},
            "inference": {"length_scale": self.length_scale, "noise_scale": self.noise_scale, "noise_w": self.noise_w},
            "vits_constants": {k: v for k, v in d.items() if k not in model_keys
                               and k not in ("sample_rate", "hop_length", "length_scale", "noise_scale", "noise_w")},
        }
        return json.dumps(out, indent=2)

    # ------------------------------------------------------------------ C struct (include/mi355vits.h)
    def to_c(self) -> "CVitsConfig":
        self.validate()
        c = CVitsConfig()
        c.num_symbols = self.num_symbols
        c.n_speakers = self.n_speakers
        c.inter_channels = self.inter_channels
        c.hidden_channels = self.hidden_channels
        c.filter_channels = self.filter_channels
        c.n_heads = self.n_heads
        c.n_layers = self.n_layers
        c.kernel_size = self.kernel_size
        c.resblock = int(self.resblock)
        c.n_resblock_kernels = len(self.resblock_kernel_sizes)
        for i, k in enumerate(self.resblock_kernel_sizes):
            c.resblock_kernel_sizes[i] = k
            c.resblock_n_dilations[i] = len(self.resblock_dilation_sizes[i])
            for j, dl in enumerate(self.resblock_dilation_sizes[i]):
                c.resblock_dilations[i * MAX_STAGES + j] = dl
        c.n_upsamples = len(self.upsample_rates)
        for i, (r, k) in enumerate(zip(self.upsample_rates, self.upsample_kernel_sizes)):
            c.upsample_rates[i] = r
            c.upsample_kernel_sizes[i] = k
        c.upsample_initial_channel = self.upsample_initial_channel
        c.gin_channels = self.gin_channels
        c.window_size = self.window_size
        c.flow_n_flows = self.flow_n_flows
        c.flow_wn_layers = self.flow_wn_layers
        c.flow_wn_kernel = self.flow_wn_kernel
        c.flow_wn_dilation_rate = self.flow_wn_dilation_rate
        c.dp_kernel_size = self.dp_kernel_size
        c.dp_dds_layers = self.dp_dds_layers
        c.dp_n_flows = self.dp_n_flows
        c.dp_num_bins = self.dp_num_bins
        c.dp_tail_bound = self.dp_tail_bound
        c.sample_rate = self.sample_rate
        c.hop_length = self.hop_length
        return c

    @staticmethod
    def from_c(c: "CVitsConfig") -> "VitsConfig":
        nk = c.n_resblock_kernels
        nu = c.n_upsamples
        return VitsConfig(
            num_symbols=c.num_symbols, n_speakers=c.n_speakers, inter_channels=c.inter_channels,
            hidden_channels=c.hidden_channels, filter_channels=c.filter_channels, n_heads=c.n_heads,
            n_layers=c.n_layers, kernel_size=c.kernel_size, resblock=str(c.resblock),
            resblock_kernel_sizes=tuple(c.resblock_kernel_sizes[i] for i in range(nk)),
            resblock_dilation_sizes=tuple(
                tuple(c.resblock_dilations[i * MAX_STAGES + j] for j in range(c.resblock_n_dilations[i]))
                for i in range(nk)),
            upsample_rates=tuple(c.upsample_rates[i] for i in range(nu)),
            upsample_initial_channel=c.upsample_initial_channel,
            upsample_kernel_sizes=tuple(c.upsample_kernel_sizes[i] for i in range(nu)),
            gin_channels=c.gin_channels, window_size=c.window_size, flow_n_flows=c.flow_n_flows,
            flow_wn_layers=c.flow_wn_layers, flow_wn_kernel=c.flow_wn_kernel,
            flow_wn_dilation_rate=c.flow_wn_dilation_rate, dp_kernel_size=c.dp_kernel_size,
            dp_dds_layers=c.dp_dds_layers, dp_n_flows=c.dp_n_flows, dp_num_bins=c.dp_num_bins,
            dp_tail_bound=c.dp_tail_bound, sample_rate=c.sample_rate, hop_length=c.hop_length,
        )


class CVitsConfig(ctypes.Structure):
    """ctypes image of ``mi355vits_config`` (include/mi355vits.h). All int32 but one float."""

    _fields_ = [
        ("num_symbols", ctypes.c_int32),
        ("n_speakers", ctypes.c_int32),
        ("inter_channels", ctypes.c_int32),
        ("hidden_channels", ctypes.c_int32),
        ("filter_channels", ctypes.c_int32),
        ("n_heads", ctypes.c_int32),
        ("n_layers", ctypes.c_int32),
        ("kernel_size", ctypes.c_int32),
        ("resblock", ctypes.c_int32),
        ("n_resblock_kernels", ctypes.c_int32),
        ("resblock_kernel_sizes", ctypes.c_int32 * MAX_STAGES),
        ("resblock_n_dilations", ctypes.c_int32 * MAX_STAGES),
        ("resblock_dilations", ctypes.c_int32 * (MAX_STAGES * MAX_STAGES)),
        ("n_upsamples", ctypes.c_int32),
        ("upsample_rates", ctypes.c_int32 * MAX_STAGES),
        ("upsample_kernel_sizes", ctypes.c_int32 * MAX_STAGES),
        ("upsample_initial_channel", ctypes.c_int32),
        ("gin_channels", ctypes.c_int32),
        ("window_size", ctypes.c_int32),
        ("flow_n_flows", ctypes.c_int32),
        ("flow_wn_layers", ctypes.c_int32),
        ("flow_wn_kernel", ctypes.c_int32),
        ("flow_wn_dilation_rate", ctypes.c_int32),
        ("dp_kernel_size", ctypes.c_int32),
        ("dp_dds_layers", ctypes.c_int32),
        ("dp_n_flows", ctypes.c_int32),
        ("dp_num_bins", ctypes.c_int32),
        ("dp_tail_bound", ctypes.c_float),
        ("sample_rate", ctypes.c_int32),
        ("hop_length", ctypes.c_int32),
    ]
