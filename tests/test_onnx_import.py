"""``generator.onnx`` importer: protobuf reader, name resolution, config inference, and the session
shim loading straight from an ``.onnx`` path.  The ONNX files come from ``torch.onnx.export`` (tests/onnx_fixture.py)
and from the hand-rolled writer (tests/onnx_writer.py)."""
import json
import os

import numpy as np
import pytest

from mimic3_amd import onnx_import as OI
from mimic3_amd import weights as W
from mimic3_amd.config import VitsConfig
from mimic3_amd.session import InferenceSession, InvalidArgument
from tests import onnx_writer as OW
from tests.onnx_fixture import export_onnx


def _variant(name):
    if name == "multispeaker":
        cfg = VitsConfig.tiny(n_speakers=3)
    elif name == "resblock1":
        cfg = VitsConfig.tiny()
        cfg.resblock = "1"
    elif name == "wide":
        cfg = VitsConfig.tiny_wide()
    else:
        cfg = VitsConfig.tiny()
    return cfg, W.synthetic_weights(cfg, seed=11)


@pytest.fixture(scope="module")
def tiny_onnx():
    cfg, w = _variant("tiny")
    return cfg, w, export_onnx(cfg, w, weight_norm_prefixes=("flow.",))


def _same_config(a: VitsConfig, b: VitsConfig):
    assert json.loads(a.to_json()) == json.loads(b.to_json())


# ------------------------------------------------------------------------------------------------ wire format
def test_reader_handles_every_tensor_encoding():
    f32 = np.arange(6, dtype=np.float32).reshape(2, 3) - 2.5
    i64 = np.array([[-3, 1 << 40]], dtype=np.int64)
    f64 = np.array([1.5, -2.25])
    blob = OW.model(
        nodes=[OW.node("Identity", ["a_raw"], ["a_alias"]),
               OW.node("Constant", [], ["c"], attrs=[OW.attr_tensor("value", OW.tensor("", f32 * 2))]),
               OW.node("Conv", ["x", "a_alias", "c"], ["y"], name="conv0",
                       attrs=[OW.attr_ints("dilations", [3]), OW.attr_int("group", 1), OW.attr_ints("strides", [2])])],
        initializers=[OW.tensor("a_raw", f32), OW.tensor("a_typed", f32, "typed", packed_dims=True),
                      OW.tensor("i", i64, "typed"), OW.tensor("d", f64, "typed"),
                      OW.tensor("ext", None, "external", dims=[4], dtype=1)],
        inputs=["x", "a_raw"], outputs=["y"])
    m = OI.parse_model(blob)
    assert m.producer == "pytorch" and m.opset == 13
    assert m.inputs == ["x"] and m.outputs == ["y"]  # initialisers listed as inputs (old IR) are not feeds
    np.testing.assert_array_equal(m.initializers["a_raw"], f32)
    np.testing.assert_array_equal(m.initializers["a_typed"], f32)
    np.testing.assert_array_equal(m.initializers["i"], i64)
    np.testing.assert_array_equal(m.initializers["d"], f64)
    assert m.initializers["ext"] is None
    conv = m.nodes[2]
    assert (conv.op, conv.name, conv.inputs) == ("Conv", "conv0", ["x", "a_alias", "c"])
    assert conv.ints == {"dilations": [3], "group": [1], "strides": [2]}
    table = OI._resolved_constants(m)
    np.testing.assert_array_equal(table["a_alias"], f32)
    np.testing.assert_array_equal(table["c"], f32 * 2)


@pytest.mark.parametrize("blob", [b"", b"not a protobuf at all", b"\x0a\xff\xff\xff\xff\x0f", b"M355VITS" + b"\0" * 64])
def test_reader_rejects_garbage(blob):
    with pytest.raises(OI.OnnxImportError):
        OI.parse_model(blob)


def test_truncated_file_is_an_error(tiny_onnx):
    _, _, blob = tiny_onnx
    with pytest.raises(OI.OnnxImportError):
        OI.parse_model(blob[: len(blob) // 2])


# ------------------------------------------------------------------------------------------------ torch exports
@pytest.mark.parametrize("variant,wn", [("tiny", ("flow.",)), ("multispeaker", ("flow.", "dec.")),
                                        ("resblock1", ("flow.", "dec.")), ("wide", ())])
def test_torch_export_round_trip(variant, wn):
    """Weights -> torch.onnx.export (weight-norm folded into anonymous initialisers) -> importer -> same weights,
    and the configuration is recovered from the graph alone."""
    cfg, w = _variant(variant)
    blob = export_onnx(cfg, w, weight_norm_prefixes=wn)
    m = OI.parse_model(blob)
    n_anon = sum(1 for n in m.initializers if n.startswith("onnx::"))
    if wn:
        assert n_anon >= cfg.flow_n_flows * cfg.flow_wn_layers * 2  # the folded WaveNet convs at least
    cfg2, t = OI.import_onnx_bytes(blob)
    _same_config(cfg, cfg2)
    assert set(t) == set(w)
    for k in w:
        assert t[k].dtype == np.float32 and t[k].shape == w[k].shape
        np.testing.assert_allclose(t[k], w[k], rtol=0, atol=2e-6, err_msg=k)  # weight-norm refold: g * v / |v|
    # and the container the engine loads is the one weights.save would have written
    cfg3, t3 = W.unpack(OI.onnx_to_m355_bytes(blob))
    _same_config(cfg, cfg3)
    np.testing.assert_array_equal(t3["dec.conv_post.weight"], t["dec.conv_post.weight"])


def test_execution_orders_cover_the_inventory():
    for cfg in (VitsConfig.apope_low(), VitsConfig.vctk_low(), VitsConfig(), VitsConfig.tiny(n_speakers=2)):
        specs = W.tensor_specs(cfg)
        convs = [n + ".weight" for n in OI.conv_execution_order(cfg)]
        points = OI.pointwise_execution_order(cfg)
        assert len(set(convs)) == len(convs) and len(set(points)) == len(points)
        biases = {n for n in specs if n.endswith(".bias")}
        assert set(convs) | set(points) | biases == set(specs)
        assert not set(convs) & set(points)


def test_sibling_and_order_passes_without_any_weight_names(tiny_onnx):
    """Strip the name of every conv weight (and of conv_post, which has no bias to hang on to)."""
    cfg, w, blob = tiny_onnx
    m = OI.parse_model(blob)
    rename = {}
    for i, n in enumerate(m.initializers):
        if n.endswith(".weight") and m.initializers[n].ndim == 3:
            rename[n] = f"onnx::Conv_{9000 + i}"
    assert any(k.endswith("dec.conv_post.weight") for k in rename)
    cfg2, t = OI.import_onnx_bytes(OW.rewrite(m, rename=rename), declared=cfg)
    _same_config(cfg, cfg2)
    for k in w:
        np.testing.assert_allclose(t[k], w[k], rtol=0, atol=2e-6, err_msg=k)


def test_missing_and_misshapen_tensors_are_loud(tiny_onnx):
    cfg, w, blob = tiny_onnx
    m = OI.parse_model(blob)
    bias = next(n for n in m.initializers if n.endswith("dec.conv_pre.bias"))
    with pytest.raises(OI.OnnxImportError, match="dec.conv_pre"):
        OI.import_onnx_bytes(OW.rewrite(m, drop={bias}), declared=cfg)
    ln = next(n for n in m.initializers if n.endswith("enc_p.emb.weight"))
    with pytest.raises(OI.OnnxImportError, match="enc_p.emb.weight"):
        OI.import_onnx_bytes(OW.rewrite(m, drop={ln}))
    wrong = VitsConfig.tiny()
    wrong.filter_channels += 8
    with pytest.raises(OI.OnnxImportError, match="filter_channels"):
        OI.import_onnx_bytes(blob, declared=wrong)


def test_external_data_is_refused(tiny_onnx):
    cfg, w, blob = tiny_onnx
    m = OI.parse_model(blob)
    name = next(n for n in m.initializers if n.endswith("dp.pre.weight"))
    m.initializers[name] = None
    # re-serialise with that tensor marked external
    inits = [OW.tensor(n, a) if a is not None else OW.tensor(n, None, "external", dims=[cfg.hidden_channels, cfg.hidden_channels, 1], dtype=1)
             for n, a in m.initializers.items()]
    nodes = [OW.node(nd.op, nd.inputs, nd.outputs, nd.name, [OW.attr_ints(k, v) for k, v in nd.ints.items()]) for nd in m.nodes]
    with pytest.raises(OI.OnnxImportError, match="external"):
        OI.import_onnx_bytes(OW.model(nodes, inits, m.inputs, m.outputs), declared=cfg)


def test_wrong_feed_names_are_refused(tiny_onnx):
    cfg, w, blob = tiny_onnx
    m = OI.parse_model(blob)
    m.inputs = ["text", "text_lengths"]
    with pytest.raises(OI.OnnxImportError, match="feed"):
        OI.import_onnx_bytes(OW.rewrite(m))


# ------------------------------------------------------------------------------------------------ voice directory
def test_convert_cli_and_config_json(tmp_path, tiny_onnx):
    cfg, w, blob = tiny_onnx
    d = tmp_path / "en_UK" / "tiny_low"
    d.mkdir(parents=True)
    (d / "generator.onnx").write_bytes(blob)
    (d / "config.json").write_text(cfg.to_json())
    assert OI.main([str(d / "generator.onnx")]) == 0
    cfg2, t = W.load(str(d / "generator.m355"))
    _same_config(cfg, cfg2)
    np.testing.assert_allclose(t["flow.flows.0.enc.in_layers.0.weight"], w["flow.flows.0.enc.in_layers.0.weight"], atol=2e-6)
    assert "sha256" in OI.describe(str(d / "generator.onnx"))
    bad = json.loads(cfg.to_json())
    bad["model"]["n_layers"] = cfg.n_layers + 1
    (d / "config.json").write_text(json.dumps(bad))
    assert OI.main([str(d / "generator.onnx"), "-o", str(d / "x.m355")]) == 1
    assert not (d / "x.m355").exists()


def test_config_json_that_relies_on_reference_defaults(tmp_path, tiny_onnx):
    """keys a config.json leaves out are not filled with this engine's "low" defaults and then held against the
    graph — the graph decides them; keys it DOES state must still agree; and a file with no model keys at all gets the
    reference's ModelConfig defaults (mimic3_tts/config.py:112-139), not ours."""
    cfg, w, blob = tiny_onnx
    d = tmp_path / "voice"
    d.mkdir()
    (d / "generator.onnx").write_bytes(blob)
    full = json.loads(cfg.to_json())
    partial = {"model": {k: full["model"][k] for k in ("num_symbols", "hidden_channels", "n_speakers")}, "audio": full["audio"],
               "inference": full["inference"]}
    (d / "config.json").write_text(json.dumps(partial))
    cfg2, t2 = OI.import_onnx(str(d / "generator.onnx"))
    assert cfg2.resblock == cfg.resblock and cfg2.upsample_rates == cfg.upsample_rates and set(t2) == set(w)
    partial["model"]["n_layers"] = cfg.n_layers + 1  # stated, and wrong
    (d / "config.json").write_text(json.dumps(partial))
    with pytest.raises(OI.OnnxImportError, match="n_layers"):
        OI.import_onnx(str(d / "generator.onnx"))
    ref_defaults = VitsConfig.from_json({"model": {"num_symbols": 60}})
    assert (ref_defaults.resblock, ref_defaults.upsample_rates, ref_defaults.upsample_initial_channel,
            ref_defaults.resblock_kernel_sizes) == ("1", (8, 8, 2, 2), 512, (3, 7, 11))
    assert ref_defaults.declared_model_keys == frozenset({"num_symbols"})


def test_session_loads_the_onnx_file_directly(emu_lib, tmp_path, tiny_onnx):
    """What Mimic 3 does: InferenceSession(str(voice_dir / "generator.onnx")) — no .m355 beside it."""
    cfg, w, blob = tiny_onnx
    d = tmp_path / "voice"
    d.mkdir()
    (d / "generator.onnx").write_bytes(blob)
    (d / "config.json").write_text(cfg.to_json())
    feed = {"input": np.array([[3, 7, 1, 9, 4]], np.int64), "input_lengths": np.array([5], np.int64),
            "scales": np.array([0.0, 1.0, 0.0], np.float32)}
    from_onnx = InferenceSession(str(d / "generator.onnx"), _library=emu_lib).run(None, feed)[0]
    from_bytes = InferenceSession(blob, _library=emu_lib).run(None, feed)[0]
    ref = InferenceSession(W.pack(cfg, w), _library=emu_lib).run(None, feed)[0]
    assert from_onnx.shape == ref.shape and from_onnx.shape[-1] > 0
    np.testing.assert_allclose(from_onnx, ref, rtol=0, atol=1e-5)  # weights equal to 2e-6 (weight-norm refold)
    np.testing.assert_array_equal(from_onnx, from_bytes)
    # a container WITHOUT a source record beside a real .onnx is ignored (nothing says it belongs to that file) ...
    W.save(str(d / "generator.m355"), cfg, w)
    s = InferenceSession(str(d / "generator.onnx"), _library=emu_lib)
    assert s._model_path.endswith(".onnx")
    # ... the one `python -m mimic3_amd.onnx_import` writes records the .onnx's size + sha256 and wins — whatever the
    # file timestamps say (the rule is explicit, no mtime comparison) ...
    assert OI.main([str(d / "generator.onnx")]) == 0
    for stamp in ((1, 1), None):
        os.utime(d / "generator.m355", stamp)
        s = InferenceSession(str(d / "generator.onnx"), _library=emu_lib)
        assert s._model_path.endswith(".m355")
        np.testing.assert_array_equal(s.run(None, feed)[0], from_onnx)
    # ... until the .onnx is replaced: the record no longer matches, so the (now broken) .onnx is what gets loaded
    (d / "generator.onnx").write_bytes(b"garbage")
    with pytest.raises(InvalidArgument, match="cannot load"):
        InferenceSession(str(d / "generator.onnx"), _library=emu_lib)
    # an empty placeholder .onnx (or none at all) always defers to the container
    (d / "generator.onnx").write_bytes(b"")
    assert InferenceSession(str(d / "generator.onnx"), _library=emu_lib)._model_path.endswith(".m355")
    os.remove(d / "generator.onnx")
    assert InferenceSession(str(d / "generator.onnx"), _library=emu_lib)._model_path.endswith(".m355")


# ------------------------------------------------------------------------------------------------ robustness
def test_reader_never_raises_anything_but_import_errors_on_corrupted_files(tiny_onnx):
    """Bit flips, truncations and random garbage: the reader either parses or raises OnnxImportError — never an
    IndexError / struct.error / MemoryError escaping to the caller (who turns it into InvalidArgument)."""
    _, _, blob = tiny_onnx
    rng = np.random.default_rng(7)
    head = bytearray(blob[:4096])
    for trial in range(300):
        b = bytearray(head)
        mode = trial % 3
        if mode == 0:
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:
            b = b[: int(rng.integers(0, len(b)))]
        else:
            b = bytearray(rng.integers(0, 256, int(rng.integers(1, 512)), dtype=np.uint8).tobytes())
        try:
            m = OI.parse_model(bytes(b))
            OI.map_tensors(m, VitsConfig.tiny())
        except OI.OnnxImportError:
            pass
    # a flipped bit in the middle of the real file: still only OnnxImportError (or a clean parse of different numbers)
    for _ in range(20):
        b = bytearray(blob)
        b[int(rng.integers(0, len(b)))] ^= 0x40
        try:
            OI.import_onnx_bytes(bytes(b))
        except OI.OnnxImportError:
            pass
