"""``generator.onnx`` -> ``generator.m355``: the format step in front of the hot path.

A Mimic 3 voice directory holds ``generator.onnx`` + ``config.json`` (``mimic3_tts/voice.py:245-376``; the model
path is ``voice_dir / "generator.onnx"``, ``voice.py:273``).  The engine wants the weights as one flat ``.m355``
container keyed by the upstream VITS state-dict names (``mimic3_amd/weights.py``).  This module reads the ONNX file
*without* the ``onnx`` package (not installed here, and not a dependency of Mimic 3 either): a ~100-line reader of the
protobuf wire format pulls out exactly what is needed — initialisers (name, dims, data) and, per node, op type,
input/output names and the integer attributes (``dilations``, ``strides``, ``group``).

Naming.  ``torch.onnx.export`` keeps ``named_parameters()`` names for plain parameters (``enc_p.emb.weight``,
``dec.conv_pre.bias``, LayerNorm ``gamma``/``beta`` ...), but every weight-normed convolution (the WaveNet layers of
the flow, HiFi-GAN unless ``remove_weight_norm()`` ran) has ``g * v / |v|`` constant-folded into an anonymous
initialiser (``onnx::Conv_1234``).  The mapping therefore runs in three passes, each verified against the shapes
``weights.tensor_specs(cfg)`` predicts:

1. exact name (after stripping one optional common prefix such as ``model_g.``);
2. sibling: a Conv/ConvTranspose node whose *bias* input was matched as ``X.bias`` gives its weight input ``X.weight``;
3. order: remaining Conv/ConvTranspose nodes, in graph (= execution) order, against the remaining conv weights in
   the order the inference graph executes them (``conv_execution_order``);
4. order, non-conv: parameters the exporter folded through a reshape (a LayerNorm ``gamma`` viewed as [1,C,1]) or a
   negation (``exp(-logs)`` of the duration predictor's ElementwiseAffine leaves ``-logs`` feeding an ``Exp``) lose
   their names too; they are matched in first-use order against ``pointwise_execution_order`` — same element
   layout required (shapes equal after dropping 1-dims), the ``-logs`` case undone explicitly.

Anything left over, any shape disagreement, weights stored outside the file or computed in-graph (export without
constant folding) raises ``OnnxImportError`` — never a silent guess.

Hyper-parameters come from ``config.json`` beside the model when present, and are cross-checked against what the
graph itself says (tensor shapes, ``strides``/``dilations`` attributes); without a ``config.json`` the graph alone
decides (``infer_config``).

Validation status: exercised on ONNX files produced in this repository — by ``torch.onnx.export`` tracing the
oracle's graph with weight-norm parametrisations attached (``tests/onnx_fixture.py``), and by a hand-rolled protobuf
writer for the corner cases.  No real ``generator.onnx`` is reachable offline, so the first contact
with a downloaded voice is still ahead; the sha256 / byte-count check of §8a-0 is what ``describe()`` prints for it.
"""
from __future__ import annotations

import hashlib
import json
import os
import struct
import sys
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from . import weights as W
from .config import VitsConfig


class OnnxImportError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------ protobuf wire format
def _varint(buf, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        if pos >= len(buf):
            raise OnnxImportError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise OnnxImportError("varint too long")


def _fields(buf) -> Iterator[Tuple[int, int, object]]:
    """Yield (field number, wire type, value) of one message; length-delimited values are memoryviews."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise OnnxImportError("truncated length-delimited field")
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise OnnxImportError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _ints(wt: int, v) -> List[int]:
    """A repeated int64 field arrives either one varint at a time or packed."""
    if wt == 0:
        return [_signed(v)]
    out = []
    pos = 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed(x))
    return out


_DTYPES = {1: "<f4", 2: "u1", 3: "i1", 5: "<i2", 6: "<i4", 7: "<i8", 9: "?", 10: "<f2", 11: "<f8", 12: "<u4", 13: "<u8"}


def _tensor(buf) -> Tuple[str, Optional[np.ndarray]]:
    """TensorProto -> (name, array).  Returns array None for tensors whose bytes live outside the file."""
    dims: List[int] = []
    dtype = 0
    name = ""
    raw = None
    floats: List[bytes] = []
    i32: List[int] = []
    i64: List[int] = []
    f64: List[bytes] = []
    external = False
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _ints(wt, v)
        elif fno == 2:
            dtype = v
        elif fno == 4:
            floats.append(bytes(v))
        elif fno == 5:
            i32 += _ints(wt, v)
        elif fno == 7:
            i64 += _ints(wt, v)
        elif fno == 8:
            name = bytes(v).decode("utf-8")
        elif fno == 9:
            raw = v
        elif fno == 10:
            f64.append(bytes(v))
        elif fno == 13 or (fno == 14 and v == 1):
            external = True
    if external:
        return name, None
    if dtype not in _DTYPES:
        raise OnnxImportError(f"tensor {name!r}: unsupported data_type {dtype}")
    dt = np.dtype(_DTYPES[dtype])
    if any(d < 0 for d in dims):
        raise OnnxImportError(f"tensor {name!r}: negative dimension in {dims}")
    count = 1
    for d in dims:
        count *= d
        if count > (1 << 40):
            raise OnnxImportError(f"tensor {name!r}: implausible dims {dims}")
    if raw is not None:
        if len(raw) != count * dt.itemsize:
            raise OnnxImportError(f"tensor {name!r}: {len(raw)} bytes of raw_data for dims {dims} of {dt}")
        arr = np.frombuffer(raw, dtype=dt, count=count)
    elif floats:
        arr = np.frombuffer(b"".join(floats), dtype="<f4")
    elif f64:
        arr = np.frombuffer(b"".join(f64), dtype="<f8")
    elif i64:
        arr = np.asarray(i64, dtype=np.int64)
    elif i32:
        arr = np.asarray(i32, dtype=np.int32).astype(dt)  # int32_data also carries int8/16, bool and fp16 bit patterns
    else:
        arr = np.zeros(0, dtype=dt)
    if arr.size != count:
        raise OnnxImportError(f"tensor {name!r}: {arr.size} elements for dims {dims}")
    return name, arr.reshape(dims)


@dataclass
class OnnxNode:
    op: str
    inputs: List[str]
    outputs: List[str]
    name: str = ""
    ints: Dict[str, List[int]] = field(default_factory=dict)   # integer / integer-list attributes
    tensor: Optional[np.ndarray] = None                        # Constant nodes: attribute "value"


def _node(buf) -> OnnxNode:
    node = OnnxNode("", [], [])
    for fno, wt, v in _fields(buf):
        if fno == 1:
            node.inputs.append(bytes(v).decode("utf-8"))
        elif fno == 2:
            node.outputs.append(bytes(v).decode("utf-8"))
        elif fno == 3:
            node.name = bytes(v).decode("utf-8")
        elif fno == 4:
            node.op = bytes(v).decode("utf-8")
        elif fno == 5:
            aname = ""
            ai: List[int] = []
            at = None
            for afno, awt, av in _fields(v):
                if afno == 1:
                    aname = bytes(av).decode("utf-8")
                elif afno == 3:
                    ai = [_signed(av)]
                elif afno == 8:
                    ai += _ints(awt, av)
                elif afno == 5:
                    at = av
            if at is not None and aname == "value":
                node.tensor = _tensor(at)[1]
            elif ai:
                node.ints[aname] = ai
    return node


@dataclass
class OnnxModel:
    initializers: "OrderedDict[str, Optional[np.ndarray]]"
    nodes: List[OnnxNode]
    inputs: List[str]
    outputs: List[str]
    producer: str = ""
    opset: int = 0
    n_bytes: int = 0
    sha256: str = ""


def parse_model(blob: bytes) -> OnnxModel:
    """ModelProto bytes -> initialisers + nodes (graph order = execution order for torch exports).  Whatever is wrong
    with the bytes, the only exception that leaves this function is OnnxImportError."""
    try:
        return _parse_model(blob)
    except OnnxImportError:
        raise
    except (ValueError, IndexError, OverflowError, MemoryError, struct.error, TypeError) as e:  # incl. UnicodeDecodeError
        raise OnnxImportError(f"not a valid ONNX protobuf: {type(e).__name__}: {e}") from None


def _parse_model(blob: bytes) -> OnnxModel:
    view = memoryview(blob)
    graph = None
    producer = ""
    opset = 0
    try:
        for fno, wt, v in _fields(view):
            if fno == 7 and wt == 2:
                graph = v
            elif fno == 2 and wt == 2:
                producer = bytes(v).decode("utf-8", "replace")
            elif fno == 8 and wt == 2:
                for ofno, owt, ov in _fields(v):
                    if ofno == 2 and owt == 0:
                        opset = max(opset, ov)
    except (IndexError, struct.error) as e:  # ran off the end of the buffer
        raise OnnxImportError(f"not a valid ONNX protobuf: {e}") from None
    if graph is None:
        raise OnnxImportError("no graph in file: not an ONNX ModelProto")
    inits: "OrderedDict[str, Optional[np.ndarray]]" = OrderedDict()
    nodes: List[OnnxNode] = []
    inputs: List[str] = []
    outputs: List[str] = []
    for fno, wt, v in _fields(graph):
        if wt != 2:
            continue
        if fno == 1:
            nodes.append(_node(v))
        elif fno == 5:
            name, arr = _tensor(v)
            inits[name] = arr
        elif fno in (11, 12):
            for vfno, vwt, vv in _fields(v):
                if vfno == 1:
                    (inputs if fno == 11 else outputs).append(bytes(vv).decode("utf-8"))
    # graph inputs that are really initialisers (old IR versions list them in both places)
    inputs = [n for n in inputs if n not in inits]
    return OnnxModel(inits, nodes, inputs, outputs, producer, opset, len(blob), hashlib.sha256(blob).hexdigest())


# ------------------------------------------------------------------------------------------------ execution order
def conv_execution_order(cfg: VitsConfig) -> List[str]:
    """Conv / ConvTranspose modules (upstream names, no ``.weight`` suffix) in the order inference executes them:
    enc_p -> duration predictor (reverse) -> flow (reverse) -> decoder."""
    gin = cfg.is_multispeaker
    out: List[str] = []
    for i in range(cfg.n_layers):
        a = f"enc_p.encoder.attn_layers.{i}"
        out += [f"{a}.conv_q", f"{a}.conv_k", f"{a}.conv_v", f"{a}.conv_o"]
        out += [f"enc_p.encoder.ffn_layers.{i}.conv_1", f"enc_p.encoder.ffn_layers.{i}.conv_2"]
    out.append("enc_p.proj")

    def dds(prefix):
        r = []
        for i in range(cfg.dp_dds_layers):
            r += [f"{prefix}.convs_sep.{i}", f"{prefix}.convs_1x1.{i}"]
        return r

    out.append("dp.pre")
    if gin:
        out.append("dp.cond")
    out += dds("dp.convs")
    out.append("dp.proj")
    for j in range(cfg.dp_n_flows - 1, 0, -1):  # reversed list with the first ConvFlow dropped
        idx = 1 + 2 * j
        out.append(f"dp.flows.{idx}.pre")
        out += dds(f"dp.flows.{idx}.convs")
        out.append(f"dp.flows.{idx}.proj")
    for j in range(cfg.flow_n_flows - 1, -1, -1):
        f = f"flow.flows.{2 * j}"
        out.append(f"{f}.pre")
        if gin:
            out.append(f"{f}.enc.cond_layer")
        for l in range(cfg.flow_wn_layers):
            out += [f"{f}.enc.in_layers.{l}", f"{f}.enc.res_skip_layers.{l}"]
        out.append(f"{f}.post")
    out.append("dec.conv_pre")
    if gin:
        out.append("dec.cond")
    nk = len(cfg.resblock_kernel_sizes)
    for i in range(len(cfg.upsample_rates)):
        out.append(f"dec.ups.{i}")
        for j in range(nk):
            n = i * nk + j
            for m in range(len(cfg.resblock_dilation_sizes[j])):
                if cfg.resblock == "2":
                    out.append(f"dec.resblocks.{n}.convs.{m}")
                else:
                    out += [f"dec.resblocks.{n}.convs1.{m}", f"dec.resblocks.{n}.convs2.{m}"]
    out.append("dec.conv_post")
    return out


def pointwise_execution_order(cfg: VitsConfig) -> List[str]:
    """The non-convolution parameters in execution order (same walk as ``conv_execution_order``)."""
    out: List[str] = ["enc_p.emb.weight"]

    def ln(name):
        return [name + ".gamma", name + ".beta"]

    for i in range(cfg.n_layers):
        a = f"enc_p.encoder.attn_layers.{i}"
        out += [f"{a}.emb_rel_k", f"{a}.emb_rel_v"]
        out += ln(f"enc_p.encoder.norm_layers_1.{i}") + ln(f"enc_p.encoder.norm_layers_2.{i}")
    if cfg.is_multispeaker:
        out.append("emb_g.weight")

    def dds(prefix):
        r = []
        for i in range(cfg.dp_dds_layers):
            r += ln(f"{prefix}.norms_1.{i}") + ln(f"{prefix}.norms_2.{i}")
        return r

    out += dds("dp.convs")
    for j in range(cfg.dp_n_flows - 1, 0, -1):
        out += dds(f"dp.flows.{1 + 2 * j}.convs")
    out += ["dp.flows.0.m", "dp.flows.0.logs"]
    return out


# ------------------------------------------------------------------------------------------------ name resolution
def _resolved_constants(model: OnnxModel) -> Dict[str, Optional[np.ndarray]]:
    """Initialisers + Constant-node outputs + Identity aliases of either (the exporter de-duplicates equal tensors
    through Identity nodes)."""
    table: Dict[str, Optional[np.ndarray]] = dict(model.initializers)
    for nd in model.nodes:
        if nd.op == "Constant" and nd.outputs and nd.tensor is not None:
            table[nd.outputs[0]] = nd.tensor
    changed = True
    while changed:
        changed = False
        for nd in model.nodes:
            if nd.op == "Identity" and nd.inputs and nd.outputs and nd.inputs[0] in table and nd.outputs[0] not in table:
                table[nd.outputs[0]] = table[nd.inputs[0]]
                changed = True
    return table


def _strip_prefix(names) -> str:
    """Longest ``xxx.`` prefix shared by every known sub-module root, e.g. ``model_g.``."""
    roots = ("enc_p.", "dp.", "flow.", "dec.", "emb_g.")
    for n in names:
        for r in roots:
            k = n.find(r)
            if k > 0 and n[k - 1] == ".":
                cand = n[:k]
                if sum(1 for m in names if m.startswith(cand)) > 10:
                    return cand
    return ""


def _named(model: OnnxModel) -> Dict[str, np.ndarray]:
    table = _resolved_constants(model)
    prefix = _strip_prefix(list(table))
    out = {}
    for n, a in table.items():
        key = n[len(prefix):] if prefix and n.startswith(prefix) else n
        out[key] = a
    return out


def _conv_nodes(model: OnnxModel) -> List[OnnxNode]:
    return [nd for nd in model.nodes if nd.op in ("Conv", "ConvTranspose") and len(nd.inputs) >= 2]


def infer_config(model: OnnxModel, base: Optional[VitsConfig] = None) -> VitsConfig:
    """Hyper-parameters from the graph itself: named tensor shapes + Conv ``strides``/``dilations`` attributes.
    ``base`` supplies what a graph cannot say (sample rate) and is overridden wherever the graph is explicit."""
    t = _named(model)
    cfg = VitsConfig() if base is None else VitsConfig.from_json(base.to_json())

    def need(name):
        if name not in t or t[name] is None:
            raise OnnxImportError(f"cannot infer the voice configuration: initialiser {name!r} not found by name "
                                  f"(give config.json explicitly)")
        return t[name]

    def count(fmt, start=0, step=1):
        n = 0
        while fmt.format(start + n * step) in t:
            n += 1
        return n

    emb = need("enc_p.emb.weight")
    cfg.num_symbols, cfg.hidden_channels = int(emb.shape[0]), int(emb.shape[1])
    cfg.n_layers = count("enc_p.encoder.attn_layers.{}.conv_q.bias")
    relk = t.get("enc_p.encoder.attn_layers.0.emb_rel_k")
    if relk is not None and relk.ndim == 3:  # otherwise n_heads / window_size stay what config.json (or the defaults) say
        cfg.window_size = (int(relk.shape[1]) - 1) // 2
        cfg.n_heads = cfg.hidden_channels // int(relk.shape[2])
    cfg.filter_channels = int(need("enc_p.encoder.ffn_layers.0.conv_1.bias").shape[0])
    if "enc_p.encoder.ffn_layers.0.conv_1.weight" in t:
        cfg.kernel_size = int(t["enc_p.encoder.ffn_layers.0.conv_1.weight"].shape[2])
    cfg.inter_channels = int(need("enc_p.proj.bias").shape[0]) // 2
    if "emb_g.weight" in t:
        cfg.n_speakers, cfg.gin_channels = (int(x) for x in t["emb_g.weight"].shape)
    else:
        cfg.n_speakers = 1
    cfg.dp_dds_layers = count("dp.convs.convs_sep.{}.bias")
    if "dp.convs.convs_sep.0.weight" in t:
        cfg.dp_kernel_size = int(t["dp.convs.convs_sep.0.weight"].shape[2])
    n_cf = count("dp.flows.{}.pre.bias", start=3, step=2)
    cfg.dp_n_flows = n_cf + 1
    cfg.dp_num_bins = (int(need("dp.flows.3.proj.bias").shape[0]) + 1) // 3
    cfg.flow_n_flows = count("flow.flows.{}.pre.bias", step=2)
    cfg.flow_wn_layers = count("flow.flows.0.enc.in_layers.{}.bias")
    cfg.upsample_initial_channel = int(need("dec.conv_pre.bias").shape[0])
    n_up = count("dec.ups.{}.bias")

    # attributes of the conv nodes, looked up through their (named) bias input
    table = _resolved_constants(model)
    prefix = _strip_prefix(list(table))
    by_bias: Dict[str, OnnxNode] = {}
    for nd in _conv_nodes(model):
        if len(nd.inputs) >= 3:
            b = nd.inputs[2]
            by_bias[b[len(prefix):] if prefix and b.startswith(prefix) else b] = nd

    def wshape(nd):
        a = table.get(nd.inputs[1])
        return None if a is None else tuple(int(x) for x in a.shape)

    nd = by_bias.get("flow.flows.0.enc.in_layers.0.bias")
    if nd is not None and wshape(nd):
        cfg.flow_wn_kernel = wshape(nd)[2]
    nd1 = by_bias.get("flow.flows.0.enc.in_layers.1.bias")
    if nd1 is not None:
        cfg.flow_wn_dilation_rate = int(nd1.ints.get("dilations", [1])[0])
    rates, ksz = [], []
    for i in range(n_up):
        nd = by_bias.get(f"dec.ups.{i}.bias")
        if nd is None or nd.op != "ConvTranspose" or not wshape(nd):
            raise OnnxImportError(f"cannot find the ConvTranspose node of dec.ups.{i}")
        rates.append(int(nd.ints.get("strides", [1])[0]))
        ksz.append(wshape(nd)[2])
    cfg.upsample_rates, cfg.upsample_kernel_sizes = tuple(rates), tuple(ksz)
    rb1 = "dec.resblocks.0.convs1.0.bias" in t
    cfg.resblock = "1" if rb1 else "2"
    stem = "convs1" if rb1 else "convs"
    n_rb = count("dec.resblocks.{}." + stem + ".0.bias")
    if n_up == 0 or n_rb % n_up:
        raise OnnxImportError(f"{n_rb} resblocks do not divide over {n_up} upsampling stages")
    nk = n_rb // n_up
    rks, rds = [], []
    for j in range(nk):
        m = 0
        dil = []
        while f"dec.resblocks.{j}.{stem}.{m}.bias" in t:
            nd = by_bias.get(f"dec.resblocks.{j}.{stem}.{m}.bias")
            if nd is None or not wshape(nd):
                raise OnnxImportError(f"cannot find the Conv node of dec.resblocks.{j}.{stem}.{m}")
            if m == 0:
                rks.append(wshape(nd)[2])
            dil.append(int(nd.ints.get("dilations", [1])[0]))
            m += 1
        rds.append(tuple(dil))
    cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes = tuple(rks), tuple(rds)
    hop = 1
    for r in rates:
        hop *= r
    cfg.hop_length = hop
    cfg.validate()
    return cfg


def map_tensors(model: OnnxModel, cfg: VitsConfig) -> Dict[str, np.ndarray]:
    """Initialisers -> upstream state-dict names (three passes, see module docstring)."""
    specs = W.tensor_specs(cfg)
    table = _resolved_constants(model)
    prefix = _strip_prefix(list(table))

    def key(n):
        return n[len(prefix):] if prefix and n.startswith(prefix) else n

    def take(spec_name, arr, how):
        if arr is None:
            raise OnnxImportError(f"{spec_name}: tensor data is stored outside the .onnx file (external data); "
                                  f"re-export with the weights embedded")
        want = tuple(specs[spec_name])
        got = tuple(int(x) for x in arr.shape)
        if got != want:
            raise OnnxImportError(f"{spec_name} ({how}): expected shape {want}, the file has {got}")
        if arr.dtype != np.float32:
            if arr.dtype not in (np.float16, np.float64):
                raise OnnxImportError(f"{spec_name}: unexpected dtype {arr.dtype}")
            arr = arr.astype(np.float32)
        out[spec_name] = np.ascontiguousarray(arr)

    out: Dict[str, np.ndarray] = {}
    used = set()
    # pass 1: by name
    for n, a in table.items():
        k = key(n)
        if k in specs and k not in out:
            take(k, a, "by name")
            used.add(n)
    # pass 2: weight through its node's bias name
    convs = _conv_nodes(model)
    for nd in convs:
        if len(nd.inputs) >= 3 and nd.inputs[1] not in used:
            b = key(nd.inputs[2])
            if b.endswith(".bias") and b in specs:
                wname = b[:-5] + ".weight"
                if wname in specs and wname not in out:
                    if nd.inputs[1] not in table:
                        raise OnnxImportError(f"{wname}: the weight of node {nd.name or nd.op!r} is computed in the "
                                              f"graph (exported without constant folding); re-export with "
                                              f"do_constant_folding=True or after remove_weight_norm()")
                    take(wname, table[nd.inputs[1]], "via bias name")
                    used.add(nd.inputs[1])
    # pass 3: execution order
    pending = [m for m in conv_execution_order(cfg) if m + ".weight" not in out]
    if pending:
        free = [nd for nd in convs if nd.inputs[1] not in used]
        # depthwise / 1x1 convs the exporter turned into something else would show up here as a count mismatch
        if len(free) < len(pending):
            raise OnnxImportError(f"{len(pending)} conv weights are unnamed ({pending[:3]}...) but only {len(free)} "
                                  f"unclaimed Conv nodes remain in the graph")
        it = iter(free)
        for mname in pending:
            want = tuple(specs[mname + ".weight"])
            for nd in it:
                a = table.get(nd.inputs[1])
                if a is not None and tuple(int(x) for x in a.shape) == want:
                    take(mname + ".weight", a, "by execution order")
                    used.add(nd.inputs[1])
                    if mname + ".bias" in specs and mname + ".bias" not in out and len(nd.inputs) >= 3:
                        take(mname + ".bias", table.get(nd.inputs[2]), "by execution order")
                        used.add(nd.inputs[2])
                    break
            else:
                raise OnnxImportError(f"{mname}.weight: no unclaimed Conv node of shape {want} left in execution order")
    # pass 4: folded pointwise parameters, first-use order
    pending = [n for n in pointwise_execution_order(cfg) if n not in out]
    if pending:
        first_use: "OrderedDict[str, str]" = OrderedDict()
        for nd in model.nodes:
            if nd.op in ("Identity", "Constant"):
                continue
            for i in nd.inputs:
                a = table.get(i)
                if i not in used and i not in first_use and a is not None and a.dtype.kind == "f" and a.size > 1:
                    first_use[i] = nd.op
        it = iter(first_use.items())

        def squeeze(shape):
            return tuple(int(x) for x in shape if int(x) != 1)

        for pname in pending:
            want = squeeze(specs[pname])
            if "emb_rel_" in pname and len(want) == 2 and want[0] == want[1]:
                raise OnnxImportError(f"{pname}: unnamed and square ({want}); a folded transpose could not be told apart")
            for cname, op in it:
                a = table[cname]
                if squeeze(a.shape) == want:
                    if pname.endswith(".logs") and op == "Exp":
                        a = -a  # exp(-logs): the exporter folded the negation into the constant
                    take(pname, a.reshape(specs[pname]), "by first-use order")
                    used.add(cname)
                    break
            else:
                raise OnnxImportError(f"{pname}: not present by name and no unclaimed constant of layout {want} "
                                      f"left in first-use order")
    missing = [n for n in specs if n not in out]
    if missing:
        raise OnnxImportError(f"{len(missing)} tensors not found in the ONNX file: {missing[:6]}"
                              f"{' ...' if len(missing) > 6 else ''}")
    return out


# ------------------------------------------------------------------------------------------------ front end
def load_voice_config(onnx_path: str, config_path: Optional[str] = None) -> Optional[VitsConfig]:
    """``config.json`` beside the model (``voice.py:256-266`` reads the same file)."""
    if config_path is None:
        cand = os.path.join(os.path.dirname(os.path.abspath(onnx_path)), "config.json")
        config_path = cand if os.path.isfile(cand) else None
    if config_path is None:
        return None
    with open(config_path, "r", encoding="utf-8") as f:
        return VitsConfig.from_json(json.load(f))


_CHECKED = ("num_symbols", "n_speakers", "hidden_channels", "inter_channels", "filter_channels", "n_heads", "n_layers",
            "kernel_size", "resblock", "resblock_kernel_sizes", "resblock_dilation_sizes", "upsample_rates",
            "upsample_kernel_sizes", "upsample_initial_channel")


def import_onnx_bytes(blob: bytes, declared: Optional[VitsConfig] = None, where: str = "<bytes>"
                      ) -> Tuple[VitsConfig, Dict[str, np.ndarray]]:
    """ONNX ModelProto bytes (+ the voice's declared config, if any) -> (config, tensors by upstream names)."""
    model = parse_model(blob)
    cfg = infer_config(model, declared)
    if declared is not None:
        present = getattr(declared, "declared_model_keys", None)
        for k in _CHECKED:
            if present is not None and k not in present:
                continue  # not stated in config.json: nothing to contradict, the graph decides
            a, b = getattr(declared, k), getattr(cfg, k)
            if k == "n_speakers":
                a, b = max(1, a), max(1, b)
            if isinstance(a, (list, tuple)):
                a, b = json.dumps(a).replace(" ", ""), json.dumps(b).replace(" ", "")
            if str(a) != str(b):
                raise OnnxImportError(f"config.json says model.{k} = {a} but the graph in {where} has {b}")
    want_inputs = {"input", "input_lengths", "scales"} | ({"sid"} if cfg.is_multispeaker else set())
    if model.inputs and not want_inputs.issubset(set(model.inputs)):
        raise OnnxImportError(f"graph inputs {model.inputs} are not the Mimic 3 feed {sorted(want_inputs)} "
                              f"(voice.py:180-218)")
    return cfg, map_tensors(model, cfg)


def import_onnx(onnx_path: str, config_path: Optional[str] = None) -> Tuple[VitsConfig, Dict[str, np.ndarray]]:
    """Read ``generator.onnx`` (+ optional ``config.json``) -> (config, tensors keyed by upstream names)."""
    with open(onnx_path, "rb") as f:
        blob = f.read()
    return import_onnx_bytes(blob, load_voice_config(onnx_path, config_path), onnx_path)


def onnx_to_m355_bytes(blob: bytes, declared: Optional[VitsConfig] = None, where: str = "<bytes>") -> bytes:
    cfg, tensors = import_onnx_bytes(blob, declared, where)
    return W.pack(cfg, tensors)


def convert(onnx_path: str, out_path: Optional[str] = None, config_path: Optional[str] = None) -> str:
    """``generator.onnx`` -> ``generator.m355`` beside it (or ``out_path``).  Returns the path written."""
    import hashlib

    with open(onnx_path, "rb") as f:
        blob = f.read()
    cfg, tensors = import_onnx_bytes(blob, load_voice_config(onnx_path, config_path), onnx_path)
    if out_path is None:
        out_path = os.path.splitext(onnx_path)[0] + ".m355"
    tmp = out_path + ".tmp%d" % os.getpid()
    # the trailer names the .onnx this container came from: the session uses the container only while they match
    W.save(tmp, cfg, tensors, source=W.source_record(len(blob), hashlib.sha256(blob).digest()))
    os.replace(tmp, out_path)
    return out_path


def describe(onnx_path: str) -> str:
    """Byte accounting of SURVEY.md §8a-0 for a downloaded voice: sha256 (compare with ``voices.json``), initialiser
    bytes, named vs anonymous tensors."""
    with open(onnx_path, "rb") as f:
        blob = f.read()
    m = parse_model(blob)
    n_bytes = sum(a.nbytes for a in m.initializers.values() if a is not None)
    anon = [n for n in m.initializers if n.startswith("onnx::") or n.isdigit()]
    lines = [f"file      {onnx_path}", f"bytes     {m.n_bytes}", f"sha256    {m.sha256}",
             f"producer  {m.producer} (opset {m.opset})", f"inputs    {m.inputs}", f"outputs   {m.outputs}",
             f"nodes     {len(m.nodes)} ({sum(1 for n in m.nodes if n.op in ('Conv', 'ConvTranspose'))} conv)",
             f"initialisers {len(m.initializers)} ({len(anon)} anonymous), {n_bytes} bytes"]
    return "\n".join(lines)


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description="Convert a Mimic 3 generator.onnx into the engine's .m355 container")
    ap.add_argument("onnx")
    ap.add_argument("-o", "--output")
    ap.add_argument("-c", "--config", help="voice config.json (default: beside the model)")
    ap.add_argument("--describe", action="store_true", help="print the byte accounting only")
    a = ap.parse_args(argv)
    if a.describe:
        print(describe(a.onnx))
        return 0
    try:
        out = convert(a.onnx, a.output, a.config)
    except OnnxImportError as e:
        print(f"error: {e}", file=sys.stderr)
        return 1
    print(out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
