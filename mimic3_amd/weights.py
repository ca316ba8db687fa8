"""Voice weights: tensor inventory, seeded synthetic initialisation and the
``.m355`` container the native engine loads.

Tensor names follow the state-dict keys of the upstream VITS generator that the
Mimic 3 trainer exports to ``generator.onnx`` (``enc_p.*``, ``dp.*``, ``flow.*``,
``dec.*``, ``emb_g.*``; weight-norm already folded), so an
ONNX-initialiser importer can map onto them later.

Container layout (little endian), parsed by ``csrc/weights_file.cpp``::

    char     magic[8]  = "M355VITS"
    uint32   version   = 1
    uint32   config_bytes            # sizeof(mi355vits_config)
    byte     config[config_bytes]    # include/mi355vits.h : mi355vits_config
    uint32   n_tensors
    repeat n_tensors:
        uint16 name_len ; char name[name_len]
        uint32 ndim     ; uint32 dims[ndim]
        uint64 offset                # byte offset into the data section (64-byte aligned)
    uint64   data_bytes
    pad to 64-byte file offset
    byte     data[data_bytes]        # float32
    optional trailer (ignored by the engine; read by ``mimic3_amd.session.resolve_voice_file``):
        char   tag[8] = "M355SRC1" ; uint64 source_bytes ; byte source_sha256[32]
        — size and sha256 of the ``generator.onnx`` this container was converted from
"""
from __future__ import annotations

import ctypes
import io
import math
import struct
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

from .config import CVitsConfig, VitsConfig

MAGIC = b"M355VITS"
VERSION = 1
SOURCE_TAG = b"M355SRC1"
SOURCE_TRAILER_BYTES = 8 + 8 + 32


def tensor_specs(cfg: VitsConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Every tensor of the inference graph with its shape."""
    cfg.validate()
    H, F, I = cfg.hidden_channels, cfg.filter_channels, cfg.inter_channels
    hd = H // cfg.n_heads
    half = I // 2
    gin = cfg.gin_channels if cfg.is_multispeaker else 0
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, cout, cin, k, bias=True):
        s[name + ".weight"] = (cout, cin, k)
        if bias:
            s[name + ".bias"] = (cout,)

    def ln(name, c):
        s[name + ".gamma"] = (c,)
        s[name + ".beta"] = (c,)

    # --- text encoder (K1-K4)
    s["enc_p.emb.weight"] = (cfg.num_symbols, H)
    for i in range(cfg.n_layers):
        a = f"enc_p.encoder.attn_layers.{i}"
        for p in ("conv_q", "conv_k", "conv_v", "conv_o"):
            conv(f"{a}.{p}", H, H, 1)
        s[f"{a}.emb_rel_k"] = (1, 2 * cfg.window_size + 1, hd)
        s[f"{a}.emb_rel_v"] = (1, 2 * cfg.window_size + 1, hd)
        ln(f"enc_p.encoder.norm_layers_1.{i}", H)
        conv(f"enc_p.encoder.ffn_layers.{i}.conv_1", F, H, cfg.kernel_size)
        conv(f"enc_p.encoder.ffn_layers.{i}.conv_2", H, F, cfg.kernel_size)
        ln(f"enc_p.encoder.norm_layers_2.{i}", H)
    conv("enc_p.proj", 2 * I, H, 1)

    # --- stochastic duration predictor, inference half only (K5)
    def dds(prefix, c):
        for i in range(cfg.dp_dds_layers):
            s[f"{prefix}.convs_sep.{i}.weight"] = (c, 1, cfg.dp_kernel_size)
            s[f"{prefix}.convs_sep.{i}.bias"] = (c,)
            conv(f"{prefix}.convs_1x1.{i}", c, c, 1)
            ln(f"{prefix}.norms_1.{i}", c)
            ln(f"{prefix}.norms_2.{i}", c)

    conv("dp.pre", H, H, 1)
    conv("dp.proj", H, H, 1)
    dds("dp.convs", H)
    if gin:
        conv("dp.cond", H, gin, 1)
    s["dp.flows.0.m"] = (2, 1)
    s["dp.flows.0.logs"] = (2, 1)
    # upstream list: [EA, CF, Flip, CF, Flip, ...] -> ConvFlows at odd indices.
    # Reverse mode drops the first ConvFlow (index 1) — it is never read, so not stored.
    for j in range(1, cfg.dp_n_flows):
        idx = 1 + 2 * j
        conv(f"dp.flows.{idx}.pre", H, 1, 1)
        dds(f"dp.flows.{idx}.convs", H)
        conv(f"dp.flows.{idx}.proj", 3 * cfg.dp_num_bins - 1, H, 1)

    # --- residual coupling flow (K8); upstream list [RCL, Flip, ...] -> RCLs at even indices
    for j in range(cfg.flow_n_flows):
        f = f"flow.flows.{2 * j}"
        conv(f"{f}.pre", H, half, 1)
        for l in range(cfg.flow_wn_layers):
            conv(f"{f}.enc.in_layers.{l}", 2 * H, H, cfg.flow_wn_kernel)
            rs = 2 * H if l < cfg.flow_wn_layers - 1 else H
            conv(f"{f}.enc.res_skip_layers.{l}", rs, H, 1)
        if gin:
            conv(f"{f}.enc.cond_layer", 2 * H * cfg.flow_wn_layers, gin, 1)
        conv(f"{f}.post", half, H, 1)

    # --- HiFi-GAN decoder (K9-K12)
    c0 = cfg.upsample_initial_channel
    conv("dec.conv_pre", c0, I, 7)
    ch = c0
    for i, (r, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        s[f"dec.ups.{i}.weight"] = (ch, ch // 2, k)  # ConvTranspose1d: [C_in, C_out, K]
        s[f"dec.ups.{i}.bias"] = (ch // 2,)
        ch //= 2
        for j, (rk, rd) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            n = i * len(cfg.resblock_kernel_sizes) + j
            if cfg.resblock == "2":
                for m in range(len(rd)):
                    conv(f"dec.resblocks.{n}.convs.{m}", ch, ch, rk)
            else:
                for m in range(len(rd)):
                    conv(f"dec.resblocks.{n}.convs1.{m}", ch, ch, rk)
                    conv(f"dec.resblocks.{n}.convs2.{m}", ch, ch, rk)
    conv("dec.conv_post", 1, ch, 7, bias=False)
    if gin:
        conv("dec.cond", c0, gin, 1)
        s["emb_g.weight"] = (cfg.n_speakers, gin)
    return s


def count_parameters(cfg: VitsConfig) -> int:
    return int(sum(int(np.prod(shape)) for shape in tensor_specs(cfg).values()))


def synthetic_weights(cfg: VitsConfig, seed: int = 1234, frames_per_id: float = 5.0) -> Dict[str, np.ndarray]:
    """Seeded random-init weights of the exact reference shapes.

    Scaled so that activations stay O(1) through the stack (fan-in normalised),
    LayerNorm affine near identity, and the duration predictor's final affine
    makes ``exp(logw)`` land around ``frames_per_id`` frames per phoneme id so the
    natural-duration path produces utterances of realistic length.
    """
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    for name, shape in tensor_specs(cfg).items():
        if name.endswith(".gamma"):
            t = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith(".beta"):
            t = 0.1 * rng.standard_normal(shape)
        elif name.endswith(".bias"):
            t = 0.05 * rng.standard_normal(shape)
        elif name == "enc_p.emb.weight":
            t = rng.standard_normal(shape) * (cfg.hidden_channels ** -0.5)
        elif name == "emb_g.weight":
            t = rng.standard_normal(shape)
        elif name.endswith("emb_rel_k") or name.endswith("emb_rel_v"):
            t = rng.standard_normal(shape) * (shape[-1] ** -0.5)
        elif name == "dp.flows.0.m":
            t = np.array([[-math.log(frames_per_id)], [0.1]])
        elif name == "dp.flows.0.logs":
            # z <- (z - m) * exp(-logs): shrink the flow output so durations vary mildly
            t = np.array([[1.2], [0.3]])
        elif name.startswith("dec.ups.") and name.endswith(".weight"):
            cin, cout, k = shape
            # each output sample sees 2 taps x cin inputs
            t = rng.standard_normal(shape) * (1.0 / math.sqrt(2 * cin))
        elif name.endswith(".weight") and len(shape) == 3:
            cout, cin, k = shape
            gain = 1.0
            if ".res_skip_layers." in name or name.endswith(".post.weight"):
                gain = 0.5
            if name.startswith("dec.resblocks."):
                gain = 0.5  # residual branch
            if ".cond_layer." in name or name.endswith("cond.weight"):
                gain = 0.3
            if name == "dec.conv_post.weight":
                gain = 0.7
            t = rng.standard_normal(shape) * (gain / math.sqrt(cin * k))
        else:
            raise AssertionError(f"no init rule for {name}")
        w[name] = np.ascontiguousarray(t, dtype=np.float32).reshape(shape)
    return w


def check_weights(cfg: VitsConfig, weights: Dict[str, np.ndarray]) -> None:
    specs = tensor_specs(cfg)
    missing = [n for n in specs if n not in weights]
    if missing:
        raise ValueError(f"missing tensors: {missing[:5]}{'...' if len(missing) > 5 else ''}")
    for n, shape in specs.items():
        got = tuple(weights[n].shape)
        if got != tuple(shape):
            raise ValueError(f"tensor {n}: expected shape {shape}, got {got}")


def source_record(size: int, sha256_digest: bytes) -> bytes:
    """Trailer naming the ``generator.onnx`` a container was converted from (size + sha256)."""
    if len(sha256_digest) != 32:
        raise ValueError("sha256 digest must be 32 bytes")
    return SOURCE_TAG + struct.pack("<Q", int(size)) + bytes(sha256_digest)


def read_source_record(path) -> "Tuple[int, bytes] | None":
    """(size, sha256) of the source ``.onnx`` recorded in the container's trailer, or None if it has none."""
    with open(path, "rb") as f:
        f.seek(0, 2)
        n = f.tell()
        if n < SOURCE_TRAILER_BYTES + 8:
            return None
        f.seek(n - SOURCE_TRAILER_BYTES)
        t = f.read(SOURCE_TRAILER_BYTES)
    if t[:8] != SOURCE_TAG:
        return None
    return struct.unpack("<Q", t[8:16])[0], t[16:48]


def pack(cfg: VitsConfig, weights: Dict[str, np.ndarray], source: "bytes | None" = None) -> bytes:
    """Serialise config + tensors into the ``.m355`` container (``source``: optional :func:`source_record`)."""
    check_weights(cfg, weights)
    specs = tensor_specs(cfg)
    c = cfg.to_c()
    cbytes = bytes(c)
    table = io.BytesIO()
    offset = 0
    blobs = []
    for name, shape in specs.items():
        arr = np.ascontiguousarray(weights[name], dtype="<f4")
        nb = name.encode("utf-8")
        table.write(struct.pack("<H", len(nb)))
        table.write(nb)
        table.write(struct.pack("<I", len(shape)))
        table.write(struct.pack(f"<{len(shape)}I", *shape))
        table.write(struct.pack("<Q", offset))
        blobs.append((offset, arr))
        offset += (arr.nbytes + 63) // 64 * 64
    head = io.BytesIO()
    head.write(MAGIC)
    head.write(struct.pack("<II", VERSION, len(cbytes)))
    head.write(cbytes)
    head.write(struct.pack("<I", len(specs)))
    head.write(table.getvalue())
    head.write(struct.pack("<Q", offset))
    hb = head.getvalue()
    pad = (-len(hb)) % 64
    data = bytearray(offset)
    for off, arr in blobs:
        data[off:off + arr.nbytes] = arr.tobytes()
    return hb + b"\0" * pad + bytes(data) + (source or b"")


def unpack(blob: bytes) -> Tuple[VitsConfig, Dict[str, np.ndarray]]:
    """Inverse of :func:`pack` (used by tests and by the oracle-side tools)."""
    if blob[:8] != MAGIC:
        raise ValueError("not an M355VITS container")
    pos = 8
    version, clen = struct.unpack_from("<II", blob, pos)
    pos += 8
    if version != VERSION:
        raise ValueError(f"unsupported container version {version}")
    if clen != ctypes.sizeof(CVitsConfig):
        raise ValueError("config struct size mismatch")
    c = CVitsConfig.from_buffer_copy(blob[pos:pos + clen])
    pos += clen
    cfg = VitsConfig.from_c(c)
    (n,) = struct.unpack_from("<I", blob, pos)
    pos += 4
    entries = []
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", blob, pos)
        pos += 2
        name = blob[pos:pos + ln].decode("utf-8")
        pos += ln
        (nd,) = struct.unpack_from("<I", blob, pos)
        pos += 4
        dims = struct.unpack_from(f"<{nd}I", blob, pos)
        pos += 4 * nd
        (off,) = struct.unpack_from("<Q", blob, pos)
        pos += 8
        entries.append((name, dims, off))
    (dbytes,) = struct.unpack_from("<Q", blob, pos)
    pos += 8
    pos += (-pos) % 64
    if len(blob) < pos + dbytes:
        raise ValueError("truncated container")
    weights = {}
    for name, dims, off in entries:
        cnt = int(np.prod(dims)) if dims else 1
        weights[name] = np.frombuffer(blob, dtype="<f4", count=cnt, offset=pos + off).reshape(dims).copy()
    return cfg, weights


def save(path, cfg: VitsConfig, weights: Dict[str, np.ndarray], source: "bytes | None" = None) -> None:
    with open(path, "wb") as f:
        f.write(pack(cfg, weights, source))


def load(path) -> Tuple[VitsConfig, Dict[str, np.ndarray]]:
    with open(path, "rb") as f:
        return unpack(f.read())
