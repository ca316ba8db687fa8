"""Test-only: a minimal ONNX (protobuf wire format) *writer*, the mirror of the reader in
``mimic3_amd/onnx_import.py``, for corner cases ``torch.onnx.export`` does not produce on demand (typed data fields
instead of ``raw_data``, unpacked repeated ints, external data, Identity-deduplicated initialisers, Constant nodes)."""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Sequence

import numpy as np


def varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def key(fno: int, wt: int) -> bytes:
    return varint((fno << 3) | wt)


def ld(fno: int, payload: bytes) -> bytes:
    return key(fno, 2) + varint(len(payload)) + payload


def vi(fno: int, v: int) -> bytes:
    return key(fno, 0) + varint(v)


_DT = {np.dtype("float32"): 1, np.dtype("int64"): 7, np.dtype("int32"): 6, np.dtype("float64"): 11, np.dtype("float16"): 10}


def tensor(name: str, arr: Optional[np.ndarray], mode: str = "raw", packed_dims: bool = False, dims=None, dtype=None) -> bytes:
    """mode: raw | typed (float_data / int64_data / double_data) | external."""
    out = b""
    shape = list(arr.shape) if arr is not None else list(dims)
    if packed_dims:
        out += ld(1, b"".join(varint(d) for d in shape))
    else:
        for d in shape:
            out += vi(1, d)
    dt = _DT[np.dtype(arr.dtype)] if arr is not None else dtype
    out += vi(2, dt)
    if mode == "raw":
        out += ld(9, np.ascontiguousarray(arr).tobytes())
    elif mode == "typed":
        if arr.dtype == np.float32:
            out += ld(4, arr.astype("<f4").tobytes())
        elif arr.dtype == np.float64:
            out += ld(10, arr.astype("<f8").tobytes())
        elif arr.dtype == np.int64:
            out += ld(7, b"".join(varint(int(x)) for x in arr.ravel()))
        else:
            raise ValueError(arr.dtype)
    elif mode == "external":
        out += ld(13, ld(1, b"location") + ld(2, b"weights.bin")) + vi(14, 1)
    out += ld(8, name.encode())
    return out


def attr_ints(name: str, values: Sequence[int]) -> bytes:
    return ld(1, name.encode()) + b"".join(vi(8, int(v)) for v in values) + vi(20, 7)


def attr_int(name: str, value: int) -> bytes:
    return ld(1, name.encode()) + vi(3, int(value)) + vi(20, 2)


def attr_tensor(name: str, t: bytes) -> bytes:
    return ld(1, name.encode()) + ld(5, t) + vi(20, 4)


def node(op: str, inputs: Sequence[str], outputs: Sequence[str], name: str = "", attrs: Sequence[bytes] = ()) -> bytes:
    out = b"".join(ld(1, i.encode()) for i in inputs) + b"".join(ld(2, o.encode()) for o in outputs)
    if name:
        out += ld(3, name.encode())
    out += ld(4, op.encode())
    out += b"".join(ld(5, a) for a in attrs)
    return out


def value_info(name: str) -> bytes:
    return ld(1, name.encode())


def model(nodes: List[bytes], initializers: List[bytes], inputs: Sequence[str], outputs: Sequence[str],
          producer: str = "pytorch", opset: int = 13) -> bytes:
    g = b"".join(ld(1, n) for n in nodes) + ld(2, b"torch_jit") + b"".join(ld(5, t) for t in initializers)
    g += b"".join(ld(11, value_info(i)) for i in inputs) + b"".join(ld(12, value_info(o)) for o in outputs)
    return vi(1, 7) + ld(2, producer.encode()) + ld(3, b"1.13") + ld(7, g) + ld(8, ld(1, b"") + vi(2, opset))


def rewrite(model_obj, drop=(), rename: Optional[Dict[str, str]] = None, modes: Optional[Dict[str, str]] = None) -> bytes:
    """Re-serialise a parsed ``OnnxModel`` (nodes keep op, inputs, outputs, int attributes) with some initialisers
    dropped / renamed / stored differently."""
    rename = rename or {}
    modes = modes or {}
    inits = []
    for n, a in model_obj.initializers.items():
        if n in drop:
            continue
        inits.append(tensor(rename.get(n, n), a, modes.get(n, "raw")))
    nodes = []
    for nd in model_obj.nodes:
        attrs = [attr_ints(k, v) if len(v) != 1 or k in ("dilations", "strides", "pads", "kernel_shape") else attr_int(k, v[0])
                 for k, v in nd.ints.items()]
        if nd.tensor is not None:
            attrs.append(attr_tensor("value", tensor("", nd.tensor)))
        nodes.append(node(nd.op, [rename.get(i, i) for i in nd.inputs], nd.outputs, nd.name, attrs))
    return model(nodes, inits, model_obj.inputs, model_obj.outputs, model_obj.producer, model_obj.opset)
