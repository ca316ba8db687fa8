"""Kernel + host logic of mimic3_amd/csrc checked on the CPU model (tests/emu) against the oracle.
These are NOT the parity tests proper (those run on the MI355X: test_gpu_parity.py); they make sure the
sources the GPU build compiles index, tile, synchronise and orchestrate correctly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mimic3_amd import weights as W
from mimic3_amd._native import Engine, NativeError
from mimic3_amd.config import VitsConfig
from oracle.vits_oracle import VitsOracle
from tests.util import check_parity, rel_rms


def test_mfma_fragment_layout_model(emu_lib):
    assert emu_lib.test_mfma_layout() == 0.0


CONV_CASES = [
    # B, Cin, Cout, T, K, dil
    (1, 2, 3, 5, 1, 1),
    (2, 16, 32, 70, 3, 1),
    (1, 32, 32, 300, 7, 12),
    (1, 6, 70, 129, 5, 2),
    (2, 64, 29, 33, 1, 1),
    (1, 96, 96, 64, 5, 1),
    (2, 384, 40, 45, 3, 1),  # deep + short: split-K kernel (encoder FFN conv_2 shape class)
    (1, 128, 64, 200, 5, 2),  # 64-channel chunks: packed LDS tile, 16-byte operand fetches
]


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d_kernels(emu_lib, impl, case):
    B, Cin, Cout, T, K, dil = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, Cout, T)).astype(np.float32)
    in_len = np.array([T] + [max(1, T - 3)] * (B - 1), np.int32)
    out_len = in_len.copy()
    y = emu_lib.test_conv1d(x, w, bias, res, dilation=dil, impl=impl, in_len=in_len, out_len=out_len, in_slope=0.1,
                            out_scale=0.5)
    tm = (torch.arange(T)[None, :] < torch.from_numpy(in_len.astype(np.int64))[:, None]).float()[:, None, :]
    xt = F.leaky_relu(torch.from_numpy(x) * tm, 0.1)
    ref = F.conv1d(xt, torch.from_numpy(w), torch.from_numpy(bias), dilation=dil, padding=(K * dil - dil) // 2)
    ref = ((ref + torch.from_numpy(res)) * 0.5 * tm).numpy()
    assert np.abs(y - ref).max() < 2e-5, np.abs(y - ref).max()


def test_conv1d_accumulate_and_ressub(emu_lib):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 8, 40)).astype(np.float32)
    w = rng.standard_normal((8, 8, 3)).astype(np.float32) * 0.2
    res = rng.standard_normal((1, 8, 40)).astype(np.float32)
    y0 = rng.standard_normal((1, 8, 40)).astype(np.float32)
    for impl in (0, 1):
        y = emu_lib.test_conv1d(x, w, None, res, impl=impl, res_sub=True, accumulate_into=y0)
        ref = y0 + (res - F.conv1d(torch.from_numpy(x), torch.from_numpy(w), padding=1).numpy())
        assert np.abs(y - ref).max() < 1e-5


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("case", [(1, 8, 4, 10, 16, 8), (2, 32, 16, 37, 8, 4), (1, 6, 3, 5, 4, 2), (1, 4, 2, 9, 3, 1), (1, 16, 8, 70, 16, 8)])
def test_conv_transpose1d(emu_lib, case, impl):
    B, Cin, Cout, Tin, K, s = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, Tin)).astype(np.float32)
    w = rng.standard_normal((Cin, Cout, K)).astype(np.float32) * 0.3
    b = rng.standard_normal(Cout).astype(np.float32)
    y = emu_lib.test_conv_transpose1d(x, w, b, s, in_slope=0.1, impl=impl)
    ref = F.conv_transpose1d(F.leaky_relu(torch.from_numpy(x), 0.1), torch.from_numpy(w), torch.from_numpy(b), stride=s,
                             padding=(K - s) // 2).numpy()
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < 1e-5


@pytest.mark.parametrize("case", [(2, 64, 32, 450, 8, 4), (1, 128, 64, 210, 16, 8), (3, 64, 16, 70, 8, 4)])
def test_conv_transpose1d_split_bf16_persistent(emu_lib, case):
    """impl 2: the polyphase upsampler on the split-bf16 staged kernel; with 64-channel chunks that is the persistent
    producer / consumer form (k_conv1d_b3_pc: 8 "CUs" in the CPU model, so workgroups loop over several tiles, one and
    two chunks per tile, a ragged last column tile, more row blocks than one).  Against fp64, and bit-identical to the
    one-tile-per-workgroup kernel (same chunk order)."""
    import os

    B, Cin, Cout, Tin, K, s = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, Tin)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(2 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    y = emu_lib.test_conv_transpose1d(x, w, b, s, in_slope=0.1, impl=2)
    ref = F.conv_transpose1d(F.leaky_relu(torch.from_numpy(x).double(), 0.1), torch.from_numpy(w).double(),
                             torch.from_numpy(b).double(), stride=s, padding=(K - s) // 2).numpy()
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < 5e-6
    y1 = emu_lib.test_conv_transpose1d(x, w, b, s, in_slope=0.1, impl=1)
    assert np.abs(y - ref).max() <= 1.5 * np.abs(y1 - ref).max() + 1e-7


def test_engine_matches_oracle_tiny_ragged(emu_lib):
    check_parity(emu_lib, VitsConfig.tiny(), B=3, Tx=11, seed=1)


def test_engine_matches_oracle_multispeaker(emu_lib):
    check_parity(emu_lib, VitsConfig.tiny(n_speakers=4), B=2, Tx=8, seed=2)


def test_engine_matches_oracle_resblock1(emu_lib):
    check_parity(emu_lib, VitsConfig.tiny(resblock="1"), B=2, Tx=7, seed=3)


def test_engine_matches_oracle_injected_noise(emu_lib):
    check_parity(emu_lib, VitsConfig.tiny(), B=2, Tx=10, seed=4, noise=True)


def test_engine_forced_durations_and_length_scale(emu_lib):
    cfg = VitsConfig.tiny()
    forced = np.full((2, 6), 3, np.int32)
    out, _ = check_parity(emu_lib, cfg, B=2, Tx=6, seed=5, forced=forced, ragged=False)
    assert list(out["lengths"]) == [6 * 3 * cfg.hop_length] * 2
    check_parity(emu_lib, cfg, B=1, Tx=6, seed=6, scales=(0.0, 1.7, 0.0))


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_fused_mrf_stage_kernel(emu_lib, math, monkeypatch):
    monkeypatch.setenv("MI355VITS_MATH", math)  # read when a handle is created: both matrix-core paths, same checks
    """Decoder stages of 64 and 32 channels go through k_mrf_fused (all resblocks of a stage in one kernel);
    compare the stage taps and the waveform with the oracle, ragged batch, and with the conv-by-conv path."""
    import os

    cfg = VitsConfig.tiny_wide()
    w = W.synthetic_weights(cfg, seed=31, frames_per_id=2.0)
    out, ora = check_parity(emu_lib, cfg, B=2, Tx=7, seed=31, weights=w)
    eng = Engine(W.pack(cfg, w), library=emu_lib)
    from tests.util import make_inputs
    ids, lengths, _ = make_inputs(cfg, 2, 7, 31)
    eng.run(ids, lengths, [0, 1, 0], debug_taps=True)
    for name in ("dec.ups.0", "dec.mrf.0", "dec.ups.1", "dec.mrf.1"):
        got, ref = eng.tap(name), ora[name]
        f = got.shape[2] // int(ora["y_lengths"].max())
        m = (np.arange(got.shape[2])[None, :] < (ora["y_lengths"] * f)[:, None])[:, None, :]
        assert rel_rms(got * m, ref * m) < 1e-5, (name, rel_rms(got * m, ref * m))
    fused_audio = eng.run(ids, lengths, [0, 1, 0])["audio"]
    eng.close()
    os.environ["MI355VITS_NO_FUSED_MRF"] = "1"
    try:
        eng2 = Engine(W.pack(cfg, w), library=emu_lib)
        plain_audio = eng2.run(ids, lengths, [0, 1, 0])["audio"]
        eng2.close()
    finally:
        del os.environ["MI355VITS_NO_FUSED_MRF"]
    for b in range(2):
        L = int(out["lengths"][b])
        assert rel_rms(fused_audio[b, :L], plain_audio[b, :L]) < 1e-5


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_partly_fused_mrf_stage_with_128_channels(emu_lib, math, monkeypatch):
    monkeypatch.setenv("MI355VITS_MATH", math)  # read when a handle is created: both matrix-core paths, same checks
    """First decoder stage of the real voices (128 channels): the two narrow resblocks (k = 3, 5) run in the fused
    kernel (4 x 2 waves, 96-column tiles), the wide one (k = 7, halo 45) conv by conv, accumulated onto the same output."""
    import os

    cfg = VitsConfig.tiny_wide(initial_channel=256)
    w = W.synthetic_weights(cfg, seed=33, frames_per_id=3.0)
    out, _ = check_parity(emu_lib, cfg, B=2, Tx=13, seed=33, weights=w)
    os.environ["MI355VITS_NO_FUSED_MRF"] = "1"
    try:
        out2, _ = check_parity(emu_lib, cfg, B=2, Tx=13, seed=33, weights=w)
    finally:
        del os.environ["MI355VITS_NO_FUSED_MRF"]
    for b in range(2):
        L = int(out["lengths"][b])
        assert rel_rms(out["audio"][b, :L], out2["audio"][b, :L]) < 1e-5


@pytest.mark.parametrize("B,Tx", [(2, 70), (1, 150)])
def test_attention_across_key_and_query_tiles(emu_lib, B, Tx):
    """Phoneme sequences longer than one 32-wide MFMA tile (and than 128: 8 key tiles in registers)."""
    check_parity(emu_lib, VitsConfig.tiny(), B=B, Tx=Tx, seed=40 + Tx, frames_per_id=1.2)


def test_single_phoneme_and_fallback_attention(emu_lib):
    check_parity(emu_lib, VitsConfig.tiny(), B=1, Tx=1, seed=51)
    check_parity(emu_lib, VitsConfig.tiny(), B=1, Tx=530, seed=52, frames_per_id=1.05)  # T > 512: VALU attention kernel


def test_generic_kernels_agree_with_mfma_path(emu_lib):
    """MI355VITS_FORCE_GENERIC=1 routes every conv / attention through the plain VALU kernels."""
    import os

    cfg = VitsConfig.tiny()
    w = W.synthetic_weights(cfg, seed=77)
    from tests.util import make_inputs
    ids, lengths, _ = make_inputs(cfg, 2, 40, 77)
    eng = Engine(W.pack(cfg, w), library=emu_lib)
    a = eng.run(ids, lengths, [0, 1, 0])
    eng.close()
    os.environ["MI355VITS_FORCE_GENERIC"] = "1"
    try:
        eng2 = Engine(W.pack(cfg, w), library=emu_lib)
        b = eng2.run(ids, lengths, [0, 1, 0])
        eng2.close()
    finally:
        del os.environ["MI355VITS_FORCE_GENERIC"]
    assert np.array_equal(a["lengths"], b["lengths"])
    for r in range(2):
        L = int(a["lengths"][r])
        assert rel_rms(a["audio"][r, :L], b["audio"][r, :L]) < 1e-5


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_fused_wavenet_layer_hidden_192(emu_lib, math, monkeypatch):
    monkeypatch.setenv("MI355VITS_MATH", math)  # read when a handle is created: both matrix-core paths, same checks
    """Flow with the real hidden width: k_wn_layer<6> (in-layer conv + gate + res/skip in one kernel, h ping-pong),
    multi-speaker conditioning included; also against the two-launch path."""
    import os

    cfg = VitsConfig.tiny_h192(n_speakers=3)
    w = W.synthetic_weights(cfg, seed=71, frames_per_id=3.0)
    out, _ = check_parity(emu_lib, cfg, B=2, Tx=14, seed=71, weights=w)
    os.environ["MI355VITS_NO_FUSED_WN"] = "1"
    try:
        out2, _ = check_parity(emu_lib, cfg, B=2, Tx=14, seed=71, weights=w)
    finally:
        del os.environ["MI355VITS_NO_FUSED_WN"]
    for b in range(2):
        L = int(out["lengths"][b])
        assert rel_rms(out["audio"][b, :L], out2["audio"][b, :L]) < 1e-5
    # the two geometries of the fused kernel (4 waves x 3 tiles for full grids, 6 waves x 2 tiles for small ones) do the
    # same arithmetic in the same order: identical bits, so the choice may depend on the batch size
    outs = {}
    for six in ("0", "1", "2"):
        os.environ["MI355VITS_WN_SIX_WAVES"] = six
        try:
            outs[six], _ = check_parity(emu_lib, cfg, B=2, Tx=14, seed=71, weights=w)
        finally:
            del os.environ["MI355VITS_WN_SIX_WAVES"]
    assert np.array_equal(outs["0"]["audio"], outs["1"]["audio"])
    assert np.array_equal(outs["0"]["audio"], outs["2"]["audio"])
    assert np.array_equal(outs["2"]["audio"], out["audio"])  # tiny grid -> the 12-wave geometry by default


def test_odd_flow_depth_folds_final_flip(emu_lib):
    cfg = VitsConfig.tiny()
    cfg.flow_n_flows = 3
    # taps of z are in physical (flipped) order for odd depth: compare audio only
    check_parity(emu_lib, cfg, B=1, Tx=6, seed=8, taps=False)


def test_batched_equals_unbatched(emu_lib):
    cfg = VitsConfig.tiny()
    w = W.synthetic_weights(cfg, seed=9)
    eng = Engine(W.pack(cfg, w), library=emu_lib)
    rng = np.random.default_rng(9)
    ids = rng.integers(1, cfg.num_symbols, size=(3, 8)).astype(np.int64)
    lengths = np.array([8, 5, 3], np.int64)
    for b in range(3):
        ids[b, lengths[b]:] = 0
    full = eng.run(ids, lengths, [0, 1, 0], want_pcm16=True)
    for b in range(3):
        one = eng.run(ids[b:b + 1, :lengths[b]], lengths[b:b + 1], [0, 1, 0], want_pcm16=True)
        L = int(one["lengths"][0])
        assert L == int(full["lengths"][b])
        assert np.array_equal(one["audio"][0, :L], full["audio"][b, :L])
        assert np.array_equal(one["pcm"][0, :L], full["pcm"][b, :L])
    eng.close()


def test_philox_noise_is_split_invariant_and_plausible(emu_lib):
    cfg = VitsConfig.tiny()
    w = W.synthetic_weights(cfg, seed=10)
    eng = Engine(W.pack(cfg, w), library=emu_lib)
    rng = np.random.default_rng(10)
    ids = rng.integers(1, cfg.num_symbols, size=(4, 8)).astype(np.int64)
    lengths = np.full(4, 8, np.int64)
    sc = [0.667, 1.0, 0.8]
    full = eng.run(ids, lengths, sc, seed=42, utterance_base=100)
    again = eng.run(ids, lengths, sc, seed=42, utterance_base=100)
    assert np.array_equal(full["audio"], again["audio"])
    half = eng.run(ids[2:], lengths[2:], sc, seed=42, utterance_base=102)
    for b in range(2):
        L = int(half["lengths"][b])
        assert L == int(full["lengths"][2 + b])
        assert np.array_equal(half["audio"][b, :L], full["audio"][2 + b, :L])
    other = eng.run(ids, lengths, sc, seed=43, utterance_base=100)
    assert not np.array_equal(other["audio"][:, :64], full["audio"][:, :64])
    eng.close()


def test_errors_are_reported_not_fatal(emu_lib):
    cfg = VitsConfig.tiny(n_speakers=3)
    eng = Engine(W.pack(cfg, W.synthetic_weights(cfg, seed=11)), library=emu_lib)
    ids = np.ones((1, 4), np.int64)
    with pytest.raises(NativeError, match="sid"):
        eng.run(ids, [4], [0, 1, 0])
    with pytest.raises(NativeError, match="phoneme id"):
        eng.run(ids * 10_000, [4], [0, 1, 0], sid=[0])
    with pytest.raises(NativeError, match="speaker id"):
        eng.run(ids, [4], [0, 1, 0], sid=[7])
    with pytest.raises(NativeError, match="input_lengths"):
        eng.run(ids, [9], [0, 1, 0], sid=[0])
    with pytest.raises(NativeError, match="length_scale"):
        eng.run(ids, [4], [0, 0.0, 0], sid=[0])
    ok = eng.run(ids, [4], [0, 1, 0], sid=[0])  # the handle is still usable
    assert ok["audio"].shape[0] == 1
    with pytest.raises(NativeError):
        Engine(b"not a container", library=emu_lib)
    blob = bytearray(W.pack(cfg, W.synthetic_weights(cfg, seed=11)))
    with pytest.raises(NativeError, match="truncated"):
        Engine(bytes(blob[: len(blob) // 2]), library=emu_lib)
    eng.close()


def test_empty_row_in_batch(emu_lib):
    """input_lengths = 0 for a row: one silent frame (y_length clamps to 1), other rows unaffected."""
    cfg = VitsConfig.tiny()
    w = W.synthetic_weights(cfg, seed=12)
    eng = Engine(W.pack(cfg, w), library=emu_lib)
    ids = np.array([[3, 4, 5, 6], [0, 0, 0, 0]], np.int64)
    out = eng.run(ids, [4, 0], [0, 1, 0])
    assert int(out["lengths"][1]) == cfg.hop_length
    solo = eng.run(ids[:1], [4], [0, 1, 0])
    L = int(solo["lengths"][0])
    assert np.array_equal(solo["audio"][0, :L], out["audio"][0, :L])
    eng.close()


@pytest.mark.parametrize("initial_channel", [128, 256])
def test_bf16x3_math_mode_matches_oracle_at_the_f32_tolerances(emu_lib, initial_channel):
    """MATH_BF16X3: the f32 operands of the dense convs split exactly into three bf16 terms, six partial products on the
    bf16 matrix cores, f32 accumulation.  Same tolerances as the f32 path (rel. RMS <= 1e-4, durations exact, every
    decoder stage) — and in fact the same error level: the split loses nothing above 2^-24."""
    cfg = VitsConfig.tiny_wide(initial_channel=initial_channel)  # fused MRF stages of 64 + 32 (and 128) channels
    w = W.synthetic_weights(cfg, seed=31, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    errs = {}
    for mode in ("f32", "bf16x3"):
        eng = Engine(blob, library=emu_lib)
        eng.set_math(mode)
        assert eng.math == mode
        out, ora = check_parity(emu_lib, cfg, B=2, Tx=9, seed=31, weights=w, engine=eng)
        errs[mode] = max(out["stage_errors"].values())
        L = int(out["lengths"][0])
        errs[mode + ".audio"] = rel_rms(out["audio"][0, :L], ora["audio"][0, 0, :L])
        lane = eng.clone()
        assert lane.math == mode  # a lane inherits the mode of the handle it was cloned from
        lane.close()
        eng.close()
    assert errs["bf16x3"] < 3 * errs["f32"] + 1e-6 and errs["bf16x3.audio"] < 3 * errs["f32.audio"] + 1e-6, errs
    with pytest.raises(Exception, match="math"):
        Engine(blob, library=emu_lib).set_math(7)


def test_bf16x3_weight_planes_are_an_exact_split():
    """pack_conv_weights_bf16x3 (restated in numpy): w == h + m + l exactly, each plane a bf16."""
    import torch

    rng = np.random.default_rng(0)
    w = (rng.standard_normal(4096) * np.exp(rng.uniform(-20, 20, 4096))).astype(np.float32)
    bf = lambda x: torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    h = bf(w); m = bf(w - h); l = bf(w - h - m)
    assert np.array_equal(h.astype(np.float64) + m + l, w.astype(np.float64))


def _conv_ref(x, w, bias, res, dil, in_len, slope, scale):
    import torch
    import torch.nn.functional as F

    B, Cin, T = x.shape
    K = w.shape[2]
    tm = (torch.arange(T)[None, :] < torch.from_numpy(in_len.astype(np.int64))[:, None]).double()[:, None, :]
    xt = F.leaky_relu(torch.from_numpy(x).double() * tm, slope)
    ref = F.conv1d(xt, torch.from_numpy(w).double(), torch.from_numpy(bias).double(), dilation=dil, padding=(K * dil - dil) // 2)
    return ((ref + torch.from_numpy(res).double()) * scale * tm).numpy()


@pytest.mark.parametrize("case", [(1, 32, 32, 600, 3, 1), (2, 64, 96, 530, 7, 3), (1, 96, 29, 700, 1, 1)])
def test_split_bf16_staged_conv_kernel(emu_lib, case):
    """k_conv1d_b3 (impl 2) against an fp64 torch conv, next to the f32-MFMA kernel (impl 1) on the same data: the
    split-operand path must be at least as accurate."""
    B, Cin, Cout, T, K, dil = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, Cout, T)).astype(np.float32)
    in_len = np.array([T] + [T - 37] * (B - 1), np.int32)
    ref = _conv_ref(x, w, bias, res, dil, in_len, 0.1, 0.5)
    err = {}
    for impl in (1, 2):
        y = emu_lib.test_conv1d(x, w, bias, res, dilation=dil, impl=impl, in_len=in_len, out_len=in_len, in_slope=0.1, out_scale=0.5)
        err[impl] = np.abs(y - ref).max()
    assert err[2] < 5e-6 and err[2] <= 1.5 * err[1] + 1e-7, err


def test_bf16x3_engine_path_long_utterance_uses_the_staged_split_kernels(emu_lib):
    """More than 512 frames: the WaveNet in-layer (gate epilogue), res/skip, flow pre/post, conv_pre, upsampler and
    resblock convs of a tiny voice all run through k_conv1d_b3 in MATH_BF16X3; parity at the f32 tolerances."""
    import os

    cfg = VitsConfig.tiny(n_speakers=3)
    w = W.synthetic_weights(cfg, seed=12, frames_per_id=2.0)
    os.environ["MI355VITS_WN_B3"] = "1"        # the tiny voice's convs are below the engine's "worth it" thresholds:
    os.environ["MI355VITS_B3_MIN_WORK"] = "0"  # route them through the split kernels anyway (read when a handle is created)
    try:
        eng = Engine(W.pack(cfg, w), library=emu_lib)
    finally:
        del os.environ["MI355VITS_WN_B3"], os.environ["MI355VITS_B3_MIN_WORK"]
    eng.set_math("bf16x3")
    Tx = 140
    forced = np.full((2, Tx), 4, np.int32)  # 560 frames
    ids = np.random.default_rng(2).integers(1, cfg.num_symbols, (2, Tx))
    out, _ = check_parity(emu_lib, cfg, ids=ids, lengths=np.array([Tx, Tx - 9]), forced=forced, noise=True, seed=12,
                          sid=np.array([0, 2]), weights=w, engine=eng)
    assert int(out["lengths"][0]) == 560 * cfg.hop_length
    eng.close()


def test_encoder_convs_on_the_slice_kernel(emu_lib):
    """The text encoder's dense convs (q/k/v, o, FFN 192 -> 768 -> 192, k = 3) in MATH_BF16X3 run on k_enc_b3: 64 x 64 tiles over
    one 192-channel slice staged once; the FFN's second conv is four slices whose raw sums the LayerNorm launch adds up.  A rule
    of the layer alone, so a row's bits do not depend on what it is batched with.  Encoder taps and waveform vs the oracle;
    batched == unbatched bitwise (rows ending inside a tile, 70 phonemes = two column tiles); and the general conv kernels
    (MI355VITS_NO_ENC_GEMM=1) stay within the same tolerance of it."""
    import os

    cfg = VitsConfig.tiny_h192()
    cfg.filter_channels = 768
    w = W.synthetic_weights(cfg, seed=41, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    ids = np.random.default_rng(8).integers(1, cfg.num_symbols, (3, 70))
    lengths = np.array([70, 33, 51])
    forced = np.full((3, 70), 2, np.int32)
    eng = Engine(blob, library=emu_lib)
    eng.set_math("bf16x3")
    eng.profile_enable(True)
    out, _ = check_parity(emu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=41, weights=w, engine=eng)
    one = eng.run(ids[1:2], lengths[1:2], (0.667, 1.0, 0.8), forced_durations=forced[1:2], seed=41, utterance_base=1)
    full = eng.run(ids, lengths, (0.667, 1.0, 0.8), forced_durations=forced, seed=41)
    L = int(one["lengths"][0])
    assert np.array_equal(full["audio"][1, :L], one["audio"][0, :L])
    eng.close()
    os.environ["MI355VITS_NO_ENC_GEMM"] = "1"
    try:
        eng = Engine(blob, library=emu_lib)
    finally:
        del os.environ["MI355VITS_NO_ENC_GEMM"]
    eng.set_math("bf16x3")
    ref = eng.run(ids, lengths, (0.667, 1.0, 0.8), forced_durations=forced, seed=41)
    eng.close()
    assert not np.array_equal(ref["audio"], full["audio"])  # two different kernels
    for b in range(3):
        Lb = int(full["lengths"][b])
        assert rel_rms(full["audio"][b, :Lb], ref["audio"][b, :Lb]) < 2e-5


def test_encoder_wide_forms_are_bitwise_the_64_column_form(emu_lib):
    """k_enc_b3w (128 columns x all of a conv's 32-row tiles per workgroup) against the 64 x 64 form, and its two ways of running a block
    of SIX row tiles (FFN conv_2, q / k / v, the couplings' pre conv: 192 output rows): one row tile per wave on six waves
    (MI355VITS_ENC_SIX8=0) or the 24 (row tile, column tile) units dealt three to a wave on eight waves (the default since round 6).
    Same products in the same order per output element: every text-side tap, z and the waveform bit for bit, ragged rows included."""
    import os

    cfg = VitsConfig.tiny_h192()
    cfg.filter_channels = 768
    w = W.synthetic_weights(cfg, seed=43, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    ids = np.random.default_rng(9).integers(1, cfg.num_symbols, (3, 150))
    lengths = np.array([150, 33, 129])
    res = {}
    for tag, env in (("narrow", {"MI355VITS_ENC_WIDE": "0"}), ("six", {"MI355VITS_ENC_WIDE": "1", "MI355VITS_ENC_SIX8": "0"}),
                     ("eight", {"MI355VITS_ENC_WIDE": "1"})):
        os.environ.update(env)
        try:
            eng = Engine(blob, library=emu_lib)
            eng.set_math("bf16x3")
            out = eng.run(ids, lengths, (0.667, 1.0, 0.8), debug_taps=True, seed=17)
            res[tag] = [eng.tap(k) for k in ("x", "stats", "dp.h", "w_ceil", "z_p", "z")] + [out["lengths"].copy(), out["audio"].copy()]
            eng.close()
        finally:
            for k in env:
                del os.environ[k]
    for tag in ("six", "eight"):
        for k, (a, b) in enumerate(zip(res["narrow"], res[tag])):
            assert np.array_equal(a, b), (tag, k)


@pytest.mark.parametrize("n_speakers", [1, 3])
def test_bf16x3_fused_wavenet_layer_kernel(emu_lib, n_speakers):
    """k_wn_layer_b3 (H = 192: 96 columns x all 384 rows per workgroup, operands split 3 x bf16, the gate in registers,
    res/skip from the u planes): flow output `z` and the waveform vs the oracle, ragged batch with rows shorter and
    longer than one 96-column tile, speaker conditioning; and against the f32 fused kernel."""
    cfg = VitsConfig.tiny_h192(n_speakers=n_speakers)
    w = W.synthetic_weights(cfg, seed=71, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    ids = np.random.default_rng(4).integers(1, cfg.num_symbols, (2, 30))
    forced = np.full((2, 30), 4, np.int32)  # 120 and 68 frames
    sid = np.array([2, 0]) if n_speakers > 1 else None
    outs = {}
    for mode in ("f32", "bf16x3"):
        eng = Engine(blob, library=emu_lib)
        eng.set_math(mode)
        outs[mode], _ = check_parity(emu_lib, cfg, ids=ids, lengths=np.array([30, 17]), forced=forced, noise=True, seed=71,
                                     sid=sid, weights=w, engine=eng)
        eng.close()
    L = int(outs["f32"]["lengths"][1])
    assert rel_rms(outs["bf16x3"]["audio"][1, :L], outs["f32"]["audio"][1, :L]) < 2e-5
    # the two tile widths of the split-bf16 layer kernel (96 columns for full grids, 32 for small ones) do the same
    # arithmetic in the same order: identical bits, so the choice may depend on the grid size
    import os

    by_nt = {}
    for nt in ("1", "3", "4"):  # (4: the 128-column form of round 6)
        os.environ["MI355VITS_WN_B3_NT"] = nt
        try:
            eng = Engine(blob, library=emu_lib)
            eng.set_math("bf16x3")
            by_nt[nt] = eng.run(ids, np.array([30, 17]), (0.667, 1.0, 0.8), sid, forced_durations=forced, seed=3)["audio"]
            eng.close()
        finally:
            del os.environ["MI355VITS_WN_B3_NT"]
    assert np.array_equal(by_nt["1"], by_nt["3"]) and np.array_equal(by_nt["4"], by_nt["3"])
    # ... and the twelve-wave form of the 96-column tile (one row tile per wave, three waves per SIMD) against the four-wave one
    by_nw = {}
    for nw in ("4", "12"):
        os.environ["MI355VITS_WN_B3_NT"] = "3"
        os.environ["MI355VITS_WN_WAVES"] = nw
        try:
            for mode in ("bf16x3", "bf16w"):
                eng = Engine(blob, library=emu_lib)
                eng.set_math(mode)
                by_nw[nw, mode] = eng.run(ids, np.array([30, 17]), (0.667, 1.0, 0.8), sid, forced_durations=forced, seed=3)["audio"]
                eng.close()
        finally:
            del os.environ["MI355VITS_WN_B3_NT"], os.environ["MI355VITS_WN_WAVES"]
    assert np.array_equal(by_nw["4", "bf16x3"], by_nt["3"])
    for mode in ("bf16x3", "bf16w"):
        assert np.array_equal(by_nw["4", mode], by_nw["12", mode]), mode
    # ... and the four-wave form's weight-fragment ring: three groups ahead (the default, round 5) against one group ahead
    os.environ["MI355VITS_WN_B3_NT"] = "3"
    os.environ["MI355VITS_WN_RING"] = "2"
    try:
        eng = Engine(blob, library=emu_lib)
        eng.set_math("bf16x3")
        ring2 = eng.run(ids, np.array([30, 17]), (0.667, 1.0, 0.8), sid, forced_durations=forced, seed=3)["audio"]
        eng.close()
    finally:
        del os.environ["MI355VITS_WN_B3_NT"], os.environ["MI355VITS_WN_RING"]
    assert np.array_equal(ring2, by_nt["3"])
    # ... and the epilogue's old-value loads three tiles ahead as buffer instructions (EP = 1), the first three issued in front of the
    # gate phase (EP = 2): the same additions on the same values
    for epi in ("1", "2"):
        os.environ["MI355VITS_WN_B3_NT"] = "3"
        os.environ["MI355VITS_WN_EPI"] = epi
        try:
            eng = Engine(blob, library=emu_lib)
            eng.set_math("bf16x3")
            got = eng.run(ids, np.array([30, 17]), (0.667, 1.0, 0.8), sid, forced_durations=forced, seed=3)["audio"]
            eng.close()
        finally:
            del os.environ["MI355VITS_WN_B3_NT"], os.environ["MI355VITS_WN_EPI"]
        assert np.array_equal(got, by_nt["3"]), epi
    # ... and the two-workgroups-per-CU form (k_wn_layer_b3_tw: 64-column tiles, 68 staged columns, the raw result gated one 32-column
    # tile at a time, B fragments single-buffered) in the default and the bf16-weights mode
    for mode in ("bf16x3", "bf16w"):
        os.environ["MI355VITS_WN_B3_NT"] = "3"
        os.environ["MI355VITS_WN_TW"] = "1"
        try:
            eng = Engine(blob, library=emu_lib)
            eng.set_math(mode)
            got = eng.run(ids, np.array([30, 17]), (0.667, 1.0, 0.8), sid, forced_durations=forced, seed=3)["audio"]
            eng.close()
        finally:
            del os.environ["MI355VITS_WN_B3_NT"], os.environ["MI355VITS_WN_TW"]
        assert np.array_equal(got, by_nw["4", mode]), mode


def test_f16x2_mode_fused_mrf_stages(emu_lib):
    """MATH_F16X2 (experimental): the fused MRF stages with both operands split into two fp16 terms (power-of-two pre-scaled,
    three products on the f16 MFMA, accumulators kept scaled by 2^17 and unscaled exactly).  Decoder stage taps and the
    waveform vs the oracle at the default tolerances, ragged batch over several workgroups, batched == unbatched bitwise;
    within f32 rounding of the bf16x3 path (and not identical to it: the mode really switches kernels)."""
    cfg = VitsConfig.tiny_wide()
    w = W.synthetic_weights(cfg, seed=83, frames_per_id=2.0)
    eng = Engine(W.pack(cfg, w), library=emu_lib)
    eng.set_math("f16x2")
    assert eng.math == "f16x2"
    Tx = 40
    forced = np.full((3, Tx), 3, np.int32)
    ids = np.random.default_rng(9).integers(1, cfg.num_symbols, (3, Tx))
    lengths = np.array([Tx, Tx - 11, 4])
    out, _ = check_parity(emu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=83, weights=w, engine=eng)
    one = eng.run(ids[1:2], lengths[1:2], (0.667, 1.0, 0.8), forced_durations=forced[1:2], seed=83, utterance_base=1)
    full = eng.run(ids, lengths, (0.667, 1.0, 0.8), forced_durations=forced, seed=83)
    L = int(one["lengths"][0])
    assert np.array_equal(full["audio"][1, :L], one["audio"][0, :L])
    eng.set_math("bf16x3")
    ref = eng.run(ids, lengths, (0.667, 1.0, 0.8), forced_durations=forced, seed=83)
    assert not np.array_equal(ref["audio"], full["audio"])
    assert rel_rms(full["audio"][0, : int(full["lengths"][0])], ref["audio"][0, : int(ref["lengths"][0])]) < 2e-5
    eng.close()
    # ... and the fused WaveNet layers (H = 192) in the same mode: flow output z and waveform vs the oracle, speaker conditioning
    cfg = VitsConfig.tiny_h192(n_speakers=3)
    w = W.synthetic_weights(cfg, seed=71, frames_per_id=2.0)
    ids = np.random.default_rng(4).integers(1, cfg.num_symbols, (2, 30))
    forced = np.full((2, 30), 4, np.int32)
    outs = {}
    for mode in ("bf16x3", "f16x2"):
        eng = Engine(W.pack(cfg, w), library=emu_lib)
        eng.set_math(mode)
        outs[mode], _ = check_parity(emu_lib, cfg, ids=ids, lengths=np.array([30, 17]), forced=forced, noise=True, seed=71,
                                     sid=np.array([2, 0]), weights=w, engine=eng)
        eng.close()
    assert not np.array_equal(outs["bf16x3"]["audio"], outs["f16x2"]["audio"])
    assert rel_rms(outs["f16x2"]["audio"][0], outs["bf16x3"]["audio"][0]) < 2e-5
    # ... and the staged convs / the persistent upsampler kernel (more than 512 frames; MI355VITS_B3_MIN_WORK=0 routes the tiny
    # voice's convs through them), with MI355VITS_F16X2_NO_CONVS as the A/B switch
    import os

    cfg = VitsConfig.tiny_wide()
    w = W.synthetic_weights(cfg, seed=12, frames_per_id=2.0)
    Tx = 140
    forced = np.full((2, Tx), 4, np.int32)
    ids = np.random.default_rng(2).integers(1, cfg.num_symbols, (2, Tx))
    res = {}
    for tag, env in (("convs", {}), ("noconvs", {"MI355VITS_F16X2_NO_CONVS": "1"})):
        os.environ["MI355VITS_B3_MIN_WORK"] = "0"
        os.environ.update(env)
        try:
            eng = Engine(W.pack(cfg, w), library=emu_lib)
        finally:
            del os.environ["MI355VITS_B3_MIN_WORK"]
            for k in env:
                del os.environ[k]
        eng.set_math("f16x2")
        res[tag], _ = check_parity(emu_lib, cfg, ids=ids, lengths=np.array([Tx, Tx - 9]), forced=forced, noise=True, seed=12, weights=w,
                                   engine=eng)
        eng.close()
    assert not np.array_equal(res["convs"]["audio"], res["noconvs"]["audio"])
    assert rel_rms(res["convs"]["audio"][0], res["noconvs"]["audio"][0]) < 2e-5


def test_bf16_weights_mode_separate_tolerance(emu_lib):
    """MATH_BF16W (BASELINE configs[4] "bf16 weights"): weights rounded to bf16 (their leading split term only), activations
    exact, f32 accumulate — a reduced-precision variant with its own tolerance (rel. RMS <= 2e-2 vs the f32 oracle), and
    clearly worse than the f32-grade modes (it must not be mistaken for them)."""
    cfg = VitsConfig.tiny_wide()
    w = W.synthetic_weights(cfg, seed=31, frames_per_id=2.0)
    from tests.util import make_inputs

    ids, lengths, _ = make_inputs(cfg, 2, 9, 31)
    ora = VitsOracle(cfg, w).infer(ids, lengths, (0.0, 1.0, 0.0))
    errs = {}
    for mode in ("bf16x3", "bf16w"):
        eng = Engine(W.pack(cfg, w), library=emu_lib)
        eng.set_math(mode)
        out = eng.run(ids, lengths, (0.0, 1.0, 0.0))
        assert np.array_equal(out["lengths"], ora["audio_lengths"])
        L = int(out["lengths"][0])
        errs[mode] = rel_rms(out["audio"][0, :L], ora["audio"][0, 0, :L])
        eng.close()
    assert errs["bf16x3"] < 1e-5 and 1e-5 < errs["bf16w"] < 2e-2, errs


@pytest.mark.parametrize("cfgname", ["tiny", "h192"])
def test_fused_dds_layer_kernel(emu_lib, cfgname, monkeypatch):
    """k_dds_layer (depthwise conv + LN + GELU + 1x1 conv on the matrix cores + LN + GELU + residual in one launch) against
    the three-launch path: same durations, logw within f32 rounding; and against the oracle (check_parity)."""
    cfg = VitsConfig.tiny(n_speakers=2) if cfgname == "tiny" else VitsConfig.tiny_h192()
    w = W.synthetic_weights(cfg, seed=91, frames_per_id=2.5)
    out, _ = check_parity(emu_lib, cfg, B=2, Tx=37, seed=91, weights=w, noise=True)
    eng = Engine(W.pack(cfg, w), library=emu_lib)
    ids = np.random.default_rng(9).integers(1, cfg.num_symbols, (2, 37))
    sid = np.array([1, 0]) if cfg.is_multispeaker else None
    eng.run(ids, [37, 20], [0, 1, 0], sid, debug_taps=True)
    fused = eng.tap("logw"), eng.tap("w_ceil")
    eng.close()
    monkeypatch.setenv("MI355VITS_NO_FUSED_DDS", "1")
    eng = Engine(W.pack(cfg, w), library=emu_lib)
    eng.run(ids, [37, 20], [0, 1, 0], sid, debug_taps=True)
    plain = eng.tap("logw"), eng.tap("w_ceil")
    eng.close()
    assert np.array_equal(fused[1], plain[1]) and np.abs(fused[0] - plain[0]).max() < 1e-4


@pytest.mark.parametrize("n_speakers", [1, 3])
def test_dds_stack_kernel(emu_lib, n_speakers, monkeypatch):
    """The duration predictor's stacks in ONE launch each (pre + the DDS layers + proj, and for a ConvFlow the spline, over a
    64-column window of x in LDS) against one launch per piece (MI355VITS_NO_DDS_STACK=1), on a ragged batch over three workgroups per
    row (row 1 ends inside the second one, row 2 inside a window's halo):
    * k_dds_stack (f32 matrix cores; MI355VITS_NO_DDS_STACK_B3=1, and the path of MATH_F32) gives the same h, logw and durations BIT
      FOR BIT — the per-element arithmetic is that of the pieces whatever the column's place in a window;
    * k_dds_stack_b3 (the default math: twelve waves, the 1x1 convs on the bf16 matrix cores with exactly split operands) agrees to
      f32 rounding with equal durations;
    and the oracle agrees with the default (check_parity)."""
    cfg = VitsConfig.tiny_h192(n_speakers=n_speakers)
    w = W.synthetic_weights(cfg, seed=93, frames_per_id=2.5)
    Tx = 75
    ids = np.random.default_rng(10).integers(1, cfg.num_symbols, (3, Tx))
    lengths = [Tx, 41, 66]
    sid = np.array([2, 0, 1]) if cfg.is_multispeaker else None
    res = {}
    monkeypatch.setenv("MI355VITS_NO_ENC_GEMM", "1")  # the pieces' 1x1 convs on the f32 matrix cores, as in the f32 stack
    for tag, env in (("stack_b3", None), ("stack_f32", "MI355VITS_NO_DDS_STACK_B3"), ("pieces", "MI355VITS_NO_DDS_STACK")):
        if env:
            monkeypatch.setenv(env, "1")
        eng = Engine(W.pack(cfg, w), library=emu_lib)
        eng.profile_enable(True)
        eng.run(ids, lengths, [0.3, 1, 0.8], sid, debug_taps=True, seed=5)
        labels = set(eng.profile_report())
        assert ("dp.stack" in labels) == (tag != "pieces") and ("convflow.stack" in labels) == (tag != "pieces"), labels
        assert ("dds.layer" in labels) == (tag == "pieces") and ("spline" in labels) == (tag == "pieces"), labels
        res[tag] = eng.tap("dp.h"), eng.tap("logw"), eng.tap("w_ceil")
        eng.close()
        if env:
            monkeypatch.delenv(env)
    for bi, L in enumerate(lengths):
        assert np.array_equal(res["stack_f32"][0][bi, :, :L], res["pieces"][0][bi, :, :L])
        assert np.array_equal(res["stack_f32"][1][bi, :, :L], res["pieces"][1][bi, :, :L])
        h_b, h_p = res["stack_b3"][0][bi, :, :L], res["pieces"][0][bi, :, :L]
        assert np.abs(h_b - h_p).max() <= 2e-5 * max(1.0, np.abs(h_p).max()), (bi, np.abs(h_b - h_p).max())
        assert np.abs(res["stack_b3"][1][bi, :, :L] - res["pieces"][1][bi, :, :L]).max() < 1e-4
    assert np.array_equal(res["stack_f32"][2], res["pieces"][2]) and np.array_equal(res["stack_b3"][2], res["pieces"][2])
    monkeypatch.delenv("MI355VITS_NO_ENC_GEMM")
    check_parity(emu_lib, cfg, ids=ids, lengths=np.array(lengths), noise=True, seed=93, weights=w, sid=sid)


@pytest.mark.parametrize("dils", [((1, 2), (2, 6), (3, 12)), ((1, 3), (1, 3), (1, 3)), ((3, 1), (2, 1), (1, 2))])
def test_bf16x3_mrf_stage_split_once_weights_in_registers(emu_lib, dils):
    """k_mrf_p (32-channel stage, MATH_BF16X3, the default path): x / x1 as bf16 planes split ONCE in LDS, the running conv's
    weight fragments in registers, v_mfma_f32_16x16x32_bf16 tiles (320 output columns per workgroup, conv1 over the extended
    range in 21 / 22 / 25 column tiles dealt to four column groups); the 64-channel stage before it runs the same kernel with two
    k-group passes per conv (96 output columns per workgroup, 4 row tiles x 2 column groups).  The "_low" voices' dilations, a narrow set and one with
    r1 > r2; ragged batch over several workgroups (row 1 ends inside a workgroup); decoder stage taps and the waveform vs the
    oracle, and vs the split-per-tap kernel (MI355VITS_NO_MRF_P=1)."""
    import os

    cfg = VitsConfig.tiny_wide()
    cfg.resblock_dilation_sizes = dils
    w = W.synthetic_weights(cfg, seed=78, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    Tx = 24
    forced = np.full((3, Tx), 4, np.int32)  # 96 frames -> 768 columns in the 32-channel stage: 3 items per row, 9 items on the
    # CPU model's 8 "CUs": the persistent loop runs more than once in both stages
    ids = np.random.default_rng(6).integers(1, cfg.num_symbols, (3, Tx))
    lengths = np.array([Tx, Tx - 9, Tx - 1])
    outs = {}
    for tag, env in (("p", None), ("fused", "MI355VITS_NO_MRF_P")):
        if env:
            os.environ[env] = "1"
        try:
            eng = Engine(blob, library=emu_lib)
            eng.set_math("bf16x3")
            eng.profile_enable(True)
            outs[tag], _ = check_parity(emu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=78, weights=w, engine=eng)
            labels = set(eng.profile_report())
            assert ("dec.mrf_p.s1" in labels) == (tag == "p") and ("dec.mrf_fused.s1" in labels) == (tag == "fused"), labels
            assert ("dec.mrf_p" in labels) == (tag == "p"), labels  # the 64-channel stage: two k-group passes per conv
            eng.close()
        finally:
            if env:
                del os.environ[env]
    for bi in range(3):
        L = int(outs["fused"]["lengths"][bi])
        assert rel_rms(outs["p"]["audio"][bi, :L], outs["fused"]["audio"][bi, :L]) < 2e-5


@pytest.mark.parametrize("dils,seg", [(((1, 2), (2, 6), (3, 12)), 96), (((1, 2), (2, 6), (3, 12)), 288), (((1, 3), (1, 3), (1, 3)), 192), (((3, 1), (2, 1), (1, 2)), 96)])
def test_mrf_row_sweep_is_bitwise_the_block_kernel(emu_lib, dils, seg, monkeypatch):
    """k_mrf_s (kernels_mrfs.cpp): the 64-channel MRF stage as a row sweep — work item = (row, segment), one pass per
    resblock, conv1 / conv2 on specialised waves with their fragments in registers for the whole segment, x / x1 planes in LDS
    rings addressed modulo their length, y accumulating the resblocks in place — against k_mrf_p on the same inputs: every decoder
    stage tap and the waveform BIT FOR BIT (same MFMA sequences, same order of additions), for the "_low" dilations, a narrow
    set, one with r1 > r2; segments of one, two and three steps' multiples; ragged batch (row 1 ends inside a segment, row 2 one
    frame short); more work items than the CPU model's 8 "CUs" (the persistent loop runs more than once)."""
    cfg = VitsConfig.tiny_wide()
    cfg.resblock_dilation_sizes = dils
    w = W.synthetic_weights(cfg, seed=78, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    Tx = 24
    forced = np.full((3, Tx), 4, np.int32)
    ids = np.random.default_rng(6).integers(1, cfg.num_symbols, (3, Tx))
    lengths = np.array([Tx, Tx - 9, Tx - 1])
    outs, taps = {}, {}
    for tag in ("p", "s"):
        if tag == "s":
            monkeypatch.setenv("MI355VITS_MRF_SWEEP_SEG", str(seg))
        else:
            monkeypatch.setenv("MI355VITS_MRF_SWEEP_SEG", "0")
        eng = Engine(blob, library=emu_lib)
        eng.set_math("bf16x3")
        eng.profile_enable(True)
        outs[tag], _ = check_parity(emu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=78, weights=w, engine=eng)
        labels = set(eng.profile_report())
        assert ("dec.mrf_s" in labels) == (tag == "s") and ("dec.mrf_p" in labels) == (tag == "p"), labels  # the 64-channel stage
        assert "dec.mrf_p.s1" in labels and "dec.mrf_s.s1" not in labels, labels  # the 32-channel stage stays on k_mrf_p
        taps[tag] = {k: eng.tap(k) for k in ("dec.mrf.0", "dec.mrf.1")}
        eng.close()
    for k in taps["p"]:
        assert np.array_equal(taps["p"][k], taps["s"][k]), k
    assert np.array_equal(outs["p"]["audio"], outs["s"]["audio"])


def test_resblock_conv_128_channels_resident_input(emu_lib, monkeypatch):
    """k_rb_conv (kernels_rbc.cpp): the six convs of the 128-channel MRF stage, each with all its input channels resident in LDS
    as two half-buffers (one refilled while the matrix cores work through the other) — against the kernels it replaces
    (k_mrf_fused for the narrow resblocks + the staged conv for k = 7: another order of summation, so within tolerance) and,
    through check_parity, against the oracle at every decoder tap; 128- and 32-column work items BIT FOR BIT (the sum order of an
    output element does not depend on the item width); ragged batch (row 1 ends inside an item, row 2 one frame short), 384
    columns per row = 3 items of 128 columns (the last one of row 1 entirely masked), 9 items on the CPU model's 8 "CUs"."""
    cfg = VitsConfig.tiny_wide(initial_channel=256)
    w = W.synthetic_weights(cfg, seed=91, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    Tx = 24
    forced = np.full((3, Tx), 4, np.int32)
    ids = np.random.default_rng(8).integers(1, cfg.num_symbols, (3, Tx))
    lengths = np.array([Tx, Tx - 9, Tx - 1])
    outs, taps = {}, {}
    # "wide" = 128-column items on the producer-wave form (k_rb_conv_pw: twelve waves, waves 8 .. 11 only stage; the default), "wide_pw0"
    # = the same items with the staging inside the matrix waves' streams (k_rb_conv), "wide_pw2" = weight fragments two steps ahead
    for tag, env in (("wide", {"MI355VITS_RBC_WIDE": "1"}), ("wide_pw0", {"MI355VITS_RBC_WIDE": "1", "MI355VITS_RBC_PW": "0"}),
                     ("wide_pw2", {"MI355VITS_RBC_WIDE": "1", "MI355VITS_RBC_PW": "2"}), ("narrow", {"MI355VITS_RBC_WIDE": "0"}),
                     ("wide_o0", {"MI355VITS_RBC_WIDE": "1", "MI355VITS_RBC_ITEM_ORDER": "0"}),  # items w, w + W, ... instead of XCD-major
                     ("narrow_o0", {"MI355VITS_RBC_WIDE": "0", "MI355VITS_RBC_ITEM_ORDER": "0"}),
                     ("old", {"MI355VITS_NO_RBC": "1"})):
        for k in ("MI355VITS_RBC_WIDE", "MI355VITS_NO_RBC", "MI355VITS_RBC_PW", "MI355VITS_RBC_ITEM_ORDER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = Engine(blob, library=emu_lib)
        eng.set_math("bf16x3")
        eng.profile_enable(True)
        outs[tag], _ = check_parity(emu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=91, weights=w, engine=eng)
        labels = set(eng.profile_report())
        assert ("dec.mrf_fused.s0" in labels) == (tag == "old"), labels
        assert "dec.rb.s0" in labels, labels
        taps[tag] = eng.tap("dec.mrf.0")
        eng.close()
    for tag in ("wide_pw0", "wide_pw2", "narrow", "wide_o0", "narrow_o0"):
        assert np.array_equal(taps["wide"], taps[tag]), tag
        assert np.array_equal(outs["wide"]["audio"], outs[tag]["audio"]), tag
    for bi in range(3):
        L = int(outs["old"]["lengths"][bi])
        L0 = L * taps["old"].shape[2] // outs["old"]["audio"].shape[1]  # the row's own columns in stage 0 (past them: unmasked leftovers)
        assert rel_rms(taps["wide"][bi, :, :L0], taps["old"][bi, :, :L0]) < 2e-6
        assert rel_rms(outs["wide"]["audio"][bi, :L], outs["old"]["audio"][bi, :L]) < 2e-5


def test_polyphase_upsamplers_resident_input(emu_lib, monkeypatch):
    """k_ups_pl / k_ups64 (kernels_rbc.cpp): the upsamplers 128 -> 64 (x 8, k 16) and 64 -> 32 (x 4, k 8) as two-tap polyphase convs
    with all input channels resident in LDS (128 -> 64: half-buffers loaded one phase ahead of their stores, phase rows in blocks of
    128; 64 -> 32: resident weights, tile-major, two whole buffers, items of 127 positions; 16- / 8-byte phase-interleaved stores) — through check_parity against the oracle at the upsampler taps, against the staged polyphase kernel
    it replaces (MI355VITS_NO_RBC=1) within tolerance, and wide vs narrow work items BIT FOR BIT; ragged rows, first / last output
    position (the half-valid phases at both ends of a row), several items per row."""
    cfg = VitsConfig.tiny_wide(initial_channel=256)
    cfg.upsample_rates = (8, 8, 4)  # the "_low" voices' decoder: 256 -> 128 -> 64 -> 32 channels
    cfg.upsample_kernel_sizes = (16, 16, 8)
    cfg.hop_length = 256
    w = W.synthetic_weights(cfg, seed=93, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    Tx = 12
    forced = np.full((2, Tx), 2, np.int32)
    ids = np.random.default_rng(9).integers(1, cfg.num_symbols, (2, Tx))
    lengths = np.array([Tx, Tx - 5])
    outs, taps = {}, {}
    # "..._st8": k_ups64 with the round-4 epilogue (two 8-byte stores per tile instead of one 16-byte store in a row's interior items)
    for tag, env in (("wide", {"MI355VITS_RBC_WIDE": "1"}), ("narrow", {"MI355VITS_RBC_WIDE": "0"}), ("old", {"MI355VITS_NO_RBC": "1"}),
                     ("wide_st8", {"MI355VITS_RBC_WIDE": "1", "MI355VITS_UPS64_ST8": "1"}), ("narrow_st8", {"MI355VITS_RBC_WIDE": "0", "MI355VITS_UPS64_ST8": "1"}),
                     ("wide_o0", {"MI355VITS_RBC_WIDE": "1", "MI355VITS_RBC_ITEM_ORDER": "0"}), ("narrow_o0", {"MI355VITS_RBC_WIDE": "0", "MI355VITS_RBC_ITEM_ORDER": "0"})):
        for k in ("MI355VITS_RBC_WIDE", "MI355VITS_NO_RBC", "MI355VITS_UPS64_ST8", "MI355VITS_RBC_ITEM_ORDER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = Engine(blob, library=emu_lib)
        eng.set_math("bf16x3")
        eng.profile_enable(True)
        outs[tag], _ = check_parity(emu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=93, weights=w, engine=eng)
        taps[tag] = {k: eng.tap(k) for k in ("dec.ups.0", "dec.ups.1", "dec.ups.2")}
        eng.close()
    for k in taps["wide"]:
        for tag in ("narrow", "wide_st8", "narrow_st8", "wide_o0", "narrow_o0"):
            assert np.array_equal(taps["wide"][k], taps[tag][k]), (k, tag)
        assert not np.array_equal(taps["wide"][k], taps["old"][k]), k  # (another kernel did run: another order of summation)
    assert np.array_equal(outs["wide"]["audio"], outs["narrow"]["audio"])
    for bi in range(2):
        L = int(outs["old"]["lengths"][bi])
        for k in taps["wide"]:
            Lk = L * taps["old"][k].shape[2] // outs["old"]["audio"].shape[1]
            assert rel_rms(taps["wide"][k][bi, :, :Lk], taps["old"][k][bi, :, :Lk]) < 2e-6, k
        assert rel_rms(outs["wide"]["audio"][bi, :L], outs["old"]["audio"][bi, :L]) < 2e-5


@pytest.mark.parametrize("kd,wide", [((3, 2), "1"), ((5, 6), "0"), ((7, 12), "1"), ((7, 3), "0")])
def test_resident_input_resblock_conv_vs_fp64(emu_lib, kd, wide, monkeypatch):
    """The kernel-level hook of k_rb_conv_pw / k_rb_conv (mi355vits_test_conv1d impl 4) on the CPU model against an fp64 conv, next to
    the f32-MFMA kernel: 128- and 32-column items, a ragged row, write and accumulate (the GPU suite runs all six shapes at
    full size: tests/test_gpu_parity.py)."""
    import torch
    import torch.nn.functional as F

    K, dil = kd
    monkeypatch.setenv("MI355VITS_RBC_WIDE", wide)
    B, T, C = 2, 300, 128
    rng = np.random.default_rng(100 * K + dil)
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32)
    res = rng.standard_normal((B, C, T)).astype(np.float32)
    y0 = rng.standard_normal((B, C, T)).astype(np.float32)
    in_len = np.array([T, T - 61], np.int32)
    tm = (torch.arange(T)[None, :] < torch.from_numpy(in_len.astype(np.int64))[:, None]).double()[:, None, :]
    xt = F.leaky_relu(torch.from_numpy(x).double() * tm, 0.1)
    conv = F.conv1d(xt, torch.from_numpy(w).double(), torch.from_numpy(bias).double(), dilation=dil, padding=(K * dil - dil) // 2)
    ref = ((conv + torch.from_numpy(res).double()) * 0.5).numpy()
    for acc in (False, True):
        want = ref + (y0.astype(np.float64) if acc else 0.0)
        err = {}
        for impl in (1, 4):
            y = emu_lib.test_conv1d(x, w, bias, res, dilation=dil, impl=impl, in_len=in_len, in_slope=0.1, out_scale=0.5,
                                    accumulate_into=y0 if acc else None)
            # over each row's own columns: items at or past a row's length are not computed (ragged batches, round 6)
            err[impl] = float(np.sqrt(np.mean(((y - want) * tm.numpy()) ** 2)))
        assert err[4] < 1e-6 and err[4] <= 1.25 * err[1] + 2e-8, (acc, err)


@pytest.mark.parametrize("case,wide", [((128, 64, 8, 16), "1"), ((64, 32, 4, 8), "1"), ((64, 32, 4, 8), "0"), ((256, 128, 8, 16), "0")])
def test_resident_input_upsamplers_vs_fp64(emu_lib, case, wide, monkeypatch):
    """mi355vits_test_conv_transpose1d impl 3 (k_ups_pl / k_ups64) on the CPU model against an fp64 ConvTranspose1d, next to the
    f32-MFMA polyphase kernel."""
    import torch
    import torch.nn.functional as F

    Cin, Cout, stride, K = case
    monkeypatch.setenv("MI355VITS_RBC_WIDE", wide)
    B, Tin = 2, 150
    rng = np.random.default_rng(Cin)
    x = rng.standard_normal((B, Cin, Tin)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(Cin * K / stride)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    ref = F.conv_transpose1d(F.leaky_relu(torch.from_numpy(x).double(), 0.1), torch.from_numpy(w).double(), torch.from_numpy(bias).double(),
                             stride=stride, padding=(K - stride) // 2).numpy()
    err = {}
    for impl in (1, 3):
        y = emu_lib.test_conv_transpose1d(x, w, bias, stride, in_slope=0.1, impl=impl)
        err[impl] = float(np.sqrt(np.mean((y - ref) ** 2)))
    assert err[3] < 1e-6 and err[3] <= 1.25 * err[1] + 2e-8, err


def test_conv_post_dpp_kernel_is_bitwise_the_round1_kernel(emu_lib, monkeypatch):
    """k_conv_post_tanh_dpp (round 5: one 16-byte load per lane, channel and tile; the taps' neighbours through DPP wave shifts; tiles of
    248 produced samples overlapping by two float4) against the round-1 kernel (MI355VITS_CONV_POST_V1=1: eight samples per lane,
    four overlapping loads per channel): the same fmaf chain in the same (channel, tap) order, so the waveform, the per-row peak
    (int16 scale) and the int16 samples must be identical — rows ending inside a tile, inside a wave's span and inside a workgroup's
    span, a one-phoneme row, and against the oracle through check_parity."""
    cfg = VitsConfig.tiny_wide(initial_channel=256)
    cfg.upsample_rates = (8, 8, 4)
    cfg.upsample_kernel_sizes = (16, 16, 8)
    cfg.hop_length = 256
    w = W.synthetic_weights(cfg, seed=95, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    Tx = 12
    forced = np.full((4, Tx), 2, np.int32)
    forced[3, :] = 1
    ids = np.random.default_rng(10).integers(1, cfg.num_symbols, (4, Tx))
    lengths = np.array([Tx, 7, 1, 9])
    outs = {}
    for tag, env in (("dpp", None), ("v1", "1")):
        if env is None:
            monkeypatch.delenv("MI355VITS_CONV_POST_V1", raising=False)
        else:
            monkeypatch.setenv("MI355VITS_CONV_POST_V1", env)
        eng = Engine(blob, library=emu_lib)
        eng.set_math("bf16x3")
        outs[tag], _ = check_parity(emu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=95, weights=w, engine=eng)
        eng.close()
    assert np.array_equal(outs["dpp"]["lengths"], outs["v1"]["lengths"])
    for b in range(4):
        L = int(outs["v1"]["lengths"][b])
        assert L > 0
        assert np.array_equal(outs["dpp"]["audio"][b, :L], outs["v1"]["audio"][b, :L]), b
        assert np.array_equal(outs["dpp"]["pcm"][b], outs["v1"]["pcm"][b]), b


ENC_CASES = [(2, 192, 576, 70, 1), (1, 192, 192, 1, 1), (2, 96, 192, 130, 1), (1, 96, 40, 65, 3), (3, 192, 768, 130, 3), (2, 768, 192, 65, 3), (1, 192, 29, 64, 1),
             (1, 384, 100, 33, 3)]


@pytest.mark.parametrize("case", ENC_CASES)
def test_encoder_slice_kernel_vs_fp64(emu_lib, case):
    """k_enc_b3 (impl 3) against an fp64 conv: pointwise and k = 3 convs of the encoder's shapes, one column, exactly one tile, one
    column into the next tile, ragged input / output masks, an output width that is not a multiple of 32, and split convs (768 and
    384 input channels: the slices' raw sums)."""
    B, Cin, Cout, T, K = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    split = Cin > 192  # (96 input channels: the half-width slice of the coupling layers' flow.pre)
    bias = None if split else rng.standard_normal(Cout).astype(np.float32)
    res = None if split else rng.standard_normal((B, Cout, T)).astype(np.float32)
    in_len = np.array([T] + [max(1, T - 5)] * (B - 1), np.int32)
    y = emu_lib.test_conv1d(x, w, bias, res, impl=3, in_len=in_len, out_len=None if split else in_len)
    tm = (torch.arange(T)[None, :] < torch.from_numpy(in_len.astype(np.int64))[:, None]).double()[:, None, :]
    ref = F.conv1d(torch.from_numpy(x).double() * tm, torch.from_numpy(w).double(), None if split else torch.from_numpy(bias).double(),
                   padding=(K - 1) // 2)
    if not split:
        ref = (ref + torch.from_numpy(res).double()) * tm
    assert np.abs(y - ref.numpy()).max() < 5e-5, np.abs(y - ref.numpy()).max()


def test_attention_with_all_operands_prefetched_is_bitwise_the_same(emu_lib, monkeypatch):
    """k_rel_attention_mfma4<NKW, 48> (head size 96: q, E_k, K, V, E_v loaded in one batch before the first product) runs the same
    MFMA sequence as the trip-by-trip form (MI355VITS_ATTN_NO_PREFETCH=1): encoder output bit for bit, T <= 128 and T <= 256."""
    cfg = VitsConfig.tiny_h192()
    w = W.synthetic_weights(cfg, seed=55, frames_per_id=1.0)
    blob = W.pack(cfg, w)
    for Tx, lengths in ((70, [70, 33]), (150, [150, 129])):
        ids = np.random.default_rng(Tx).integers(1, cfg.num_symbols, (2, Tx))
        taps = {}
        for tag in ("pre", "trips"):
            if tag == "trips":
                monkeypatch.setenv("MI355VITS_ATTN_NO_PREFETCH", "1")
            eng = Engine(blob, library=emu_lib)
            eng.run(ids, lengths, [0, 1, 0], debug_taps=True)
            taps[tag] = eng.tap("x")
            eng.close()
        monkeypatch.delenv("MI355VITS_ATTN_NO_PREFETCH")
        assert np.array_equal(taps["pre"], taps["trips"])


def test_o_proj_residual_layernorm_in_one_launch(emu_lib, monkeypatch):
    """k_enc_o_ln (o-proj + residual + LayerNorm, 192 channels, in place on x) vs the two launches (MI355VITS_NO_ENC_O_LN=1): the
    encoder output to f32 rounding (the LayerNorm sums 12 groups of 16 channels instead of 16 of 12), batched == unbatched bit for
    bit, rows ending inside a 32-column tile; labels show which path ran; the oracle agrees (check_parity)."""
    cfg = VitsConfig.tiny_h192()
    cfg.n_layers = 2
    w = W.synthetic_weights(cfg, seed=57, frames_per_id=1.0)
    blob = W.pack(cfg, w)
    ids = np.random.default_rng(11).integers(1, cfg.num_symbols, (3, 70))
    lengths = [70, 33, 1]
    taps = {}
    for tag in ("fused", "two"):
        if tag == "two":
            monkeypatch.setenv("MI355VITS_NO_ENC_O_LN", "1")
        eng = Engine(blob, library=emu_lib)
        eng.profile_enable(True)
        eng.run(ids, lengths, [0, 1, 0], debug_taps=True)
        labels = set(eng.profile_report())
        assert ("enc.o_ln" in labels) == (tag == "fused") and ("enc.o" in labels) == (tag == "two"), labels
        taps[tag] = eng.tap("x")
        if tag == "fused":
            one = eng.run(ids[1:2], lengths[1:2], [0, 1, 0], debug_taps=True, utterance_base=1)
            x1 = eng.tap("x")
            assert np.array_equal(x1[0, :, :33], taps["fused"][1, :, :33])
        eng.close()
    monkeypatch.delenv("MI355VITS_NO_ENC_O_LN")
    for bi, L in enumerate(lengths):
        assert np.abs(taps["fused"][bi, :, :L] - taps["two"][bi, :, :L]).max() < 2e-5
    check_parity(emu_lib, cfg, ids=ids, lengths=np.array(lengths), noise=True, seed=57, weights=w)


def test_flow_pointwise_convs_on_the_slice_kernel(emu_lib, monkeypatch):
    """flow.pre (96 -> 192, the 96-channel slice form) and flow.post (192 -> 96, with the coupling's x1 - m epilogue) at frame
    resolution on k_enc_b3 vs the general conv kernels (MI355VITS_NO_FLOW_GEMM=1): z to f32 rounding, batched == unbatched bit for bit
    (frames depend on what a row is batched with; the kernel and its tile are fixed by the layer), oracle parity."""
    cfg = VitsConfig.tiny_h192()
    cfg.inter_channels = 192
    w = W.synthetic_weights(cfg, seed=59, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    ids = np.random.default_rng(12).integers(1, cfg.num_symbols, (3, 40))
    lengths = [40, 17, 33]
    forced = np.full((3, 40), 2, np.int32)
    forced[1, :] = 3
    zs = {}
    for tag in ("slice", "general"):
        if tag == "general":
            monkeypatch.setenv("MI355VITS_NO_FLOW_GEMM", "1")
        eng = Engine(blob, library=emu_lib)
        full = eng.run(ids, lengths, [0.667, 1.0, 0.8], forced_durations=forced, seed=3, debug_taps=True)
        zs[tag] = eng.tap("z"), full["lengths"].copy()
        if tag == "slice":
            one = eng.run(ids[1:2], lengths[1:2], [0.667, 1.0, 0.8], forced_durations=forced[1:2], seed=3, utterance_base=1)
            L = int(one["lengths"][0])
            assert np.array_equal(full["audio"][1, :L], one["audio"][0, :L])
        eng.close()
    monkeypatch.delenv("MI355VITS_NO_FLOW_GEMM")
    assert np.array_equal(zs["slice"][1], zs["general"][1])
    for bi in range(3):
        n = int(zs["slice"][1][bi]) // cfg.hop_length
        a, b = zs["slice"][0][bi, :, :n], zs["general"][0][bi, :, :n]
        assert np.abs(a - b).max() <= 5e-5 * max(1.0, np.abs(b).max())
    check_parity(emu_lib, cfg, ids=ids, lengths=np.array(lengths), forced=forced, noise=True, seed=59, weights=w)


@pytest.mark.parametrize("Tx", [1, 64, 65, 257])
def test_sequence_length_extremes_at_the_real_hidden_width(emu_lib, Tx):
    """The 192-channel text-side kernels at one phoneme, exactly / one past a 64-column tile, and past 256 phonemes (the attention
    kernel without prefetch); ragged second row; noise on."""
    cfg = VitsConfig.tiny_h192()
    rng = np.random.default_rng(70 + Tx)
    ids = rng.integers(1, cfg.num_symbols, (2, Tx))
    lengths = np.array([Tx, max(1, Tx - 7)])
    check_parity(emu_lib, cfg, ids=ids, lengths=lengths, noise=True, seed=70 + Tx, frames_per_id=1.1)


def test_bf16_weights_mode_keeps_the_text_side_exact(emu_lib):
    """MATH_BF16W rounds the weights of the frame-rate convs only (flow, decoder).  The text side — encoder, duration predictor —
    runs the exact three-term split in this mode too: ceil(exp(logw) * length_scale) is discontinuous, so a rounded duration
    predictor could change utterance lengths.  At NATURAL durations: encoder output, prior statistics, durations and lengths are
    bitwise those of MATH_BF16X3; the waveform is at the reduced-precision tolerance and clearly worse than MATH_BF16X3's."""
    cfg = VitsConfig.tiny_h192()
    cfg.filter_channels = 768
    w = W.synthetic_weights(cfg, seed=61, frames_per_id=2.0)
    ids = np.random.default_rng(13).integers(1, cfg.num_symbols, (2, 70))
    lengths = np.array([70, 41])
    ora = VitsOracle(cfg, w).infer(ids, lengths, (0.0, 1.0, 0.0))
    got = {}
    for mode in ("bf16x3", "bf16w"):
        eng = Engine(W.pack(cfg, w), library=emu_lib)
        eng.set_math(mode)
        eng.profile_enable(True)
        out = eng.run(ids, lengths, (0.0, 1.0, 0.0), debug_taps=True)
        assert {"enc.o_ln", "dp.stack", "enc.ffn2"} <= set(eng.profile_report())
        assert np.array_equal(out["lengths"], ora["audio_lengths"])
        L = int(out["lengths"][0])
        got[mode] = dict(x=eng.tap("x"), stats=eng.tap("stats"), w_ceil=eng.tap("w_ceil"), lengths=out["lengths"].copy(),
                         err=rel_rms(out["audio"][0, :L], ora["audio"][0, 0, :L]))
        eng.close()
    for k in ("x", "stats", "w_ceil", "lengths"):
        assert np.array_equal(got["bf16x3"][k], got["bf16w"][k]), k
    assert np.array_equal(got["bf16w"]["w_ceil"], ora["w_ceil"])
    assert got["bf16x3"]["err"] < 1e-5 and 1e-5 < got["bf16w"]["err"] < 2e-2, got


def test_a_dropped_partial_product_is_caught(emu_lib, monkeypatch):
    """the suite must fail when a split kernel loses one of its six partial products.  The CPU model's b3 loops (the
    WaveNet layers, the staged convs, the encoder convs, the duration-predictor stacks) can leave out one of the three SMALL products on
    request (MI355VITS_EMU_DROP = 1: l_w x h_x, 2: h_w x l_x, 3: m_w x m_x; csrc/b3.h, compiled into the CPU model only).  Measured here:
    the intact engine is 3.3e-7 from the fp64 oracle (PyTorch fp32 itself: 3.6e-7); without l_w x h_x 2.0e-6, without m x m 6.1e-6,
    without h_w x l_x 1.2e-5 — all three inside the CONTRACT tolerance (1e-4), which is exactly why the contract figure is not what the
    suite checks.  Two guards catch them:
      * check_parity's bound for the f32-grade modes (5e-6 vs the fp32 oracle) trips for the two larger ones;
      * the self-calibrating criterion `error vs fp64 <= 3 x (PyTorch fp32's own error vs fp64)` (tests/util.py f32_grade_vs_fp64) trips
        for all three, the weights' rounded l plane included, and holds for the intact engine in both f32-grade modes.
    Coverage of the injection itself: b3_chunk (WaveNet in-layer / res-skip convs, staged convs, encoder slices, DDS stacks).  The
    lean / two-term loops (b3_chunk_lean of the polyphase upsamplers, h2_chunk of MATH_F16X2) and the MRF kernels' own MFMA sweeps
    (k_mrf_p / k_mrf_s / k_rb_conv) carry no injection point: for them the guard is exercised by the fp64 kernel-level tests only."""
    import torch

    from tests.util import REL_RMS_TOL, TIGHT_REL_RMS_TOL, f32_grade_vs_fp64

    cfg = VitsConfig.tiny_h192()
    w = W.synthetic_weights(cfg, seed=71, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    ids = np.random.default_rng(4).integers(1, cfg.num_symbols, (2, 30))
    lengths = np.array([30, 17])
    forced = np.full((2, 30), 4, np.int32)
    rng = np.random.default_rng(1)
    nw = rng.standard_normal((2, 2, 30)).astype(np.float32)
    nz = rng.standard_normal((2, cfg.inter_channels, 120)).astype(np.float32)
    scales = (0.667, 1.0, 0.8)
    o64 = VitsOracle(cfg, w, dtype=torch.float64).infer(ids, lengths, scales, noise_w=nw, noise_z=nz, forced_durations=forced)
    o32 = VitsOracle(cfg, w).infer(ids, lengths, scales, noise_w=nw, noise_z=nz, forced_durations=forced)

    def engine_audio(mode, drop):
        if drop:
            monkeypatch.setenv("MI355VITS_EMU_DROP", str(drop))
        else:
            monkeypatch.delenv("MI355VITS_EMU_DROP", raising=False)
        eng = Engine(blob, library=emu_lib)
        eng.set_math(mode)
        out = eng.run(ids, lengths, scales, forced_durations=forced, noise_w=nw, noise_z=nz)
        eng.close()
        monkeypatch.delenv("MI355VITS_EMU_DROP", raising=False)
        return out

    for mode in ("bf16x3", "f32"):
        ok, e, e32 = f32_grade_vs_fp64(engine_audio(mode, 0), o64, o32)
        assert ok, (mode, e, e32)
    intact = engine_audio("bf16x3", 0)
    caught_by_tight = 0
    for drop in (1, 2, 3):
        out = engine_audio("bf16x3", drop)
        ok, e, e32 = f32_grade_vs_fp64(out, o64, o32)
        assert not ok, (drop, e, e32)                  # the self-calibrating guard sees every one of them
        assert e < REL_RMS_TOL, (drop, e)              # ... although all are inside the contract tolerance
        worst = max(rel_rms(out["audio"][b, :int(out["lengths"][b])], intact["audio"][b, :int(out["lengths"][b])]) for b in range(2))
        caught_by_tight += worst > TIGHT_REL_RMS_TOL
    assert caught_by_tight >= 2                        # check_parity's 5e-6 bound: h_w x l_x and m x m


def test_ragged_batch_computes_only_the_rows_own_items(emu_lib):
    """Round 6: the persistent decoder / flow kernels walk the (row, column block) items that HAVE work — numbered row by row, a
    cursor per workgroup, the XCD eighths equal in valid items — instead of rows x the longest row.  Very unequal rows (1 .. 40 ids,
    several rows shorter than one work item, the longest over many): every row bitwise what it is alone, all taps and the waveform
    against the oracle, and again on a REUSED handle whose workspace still holds the previous, longer batch (columns past a row's end
    keep stale values: they must never reach a valid sample)."""
    cfg = VitsConfig.tiny_wide(initial_channel=256)
    w = W.synthetic_weights(cfg, seed=91, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    Tx = 40
    rng = np.random.default_rng(12)
    ids = rng.integers(1, cfg.num_symbols, (6, Tx))
    lengths = np.array([Tx, 3, 22, 1, Tx, 9])
    forced = np.full((6, Tx), 4, np.int32)
    eng = Engine(blob, library=emu_lib)
    eng.set_math("bf16x3")
    big = eng.run(rng.integers(1, cfg.num_symbols, (6, Tx)), np.full(6, Tx), [0.667, 1.0, 0.8], forced_durations=forced, seed=1)  # fills the workspace
    assert np.isfinite(big["audio"]).all()
    out, _ = check_parity(emu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=12, weights=w, engine=eng)
    rng2 = np.random.default_rng(12 + 7)  # check_parity's noise draws (tests/util.py): the same tensors for the solo runs
    nw = rng2.standard_normal((6, 2, Tx)).astype(np.float32)
    nz = rng2.standard_normal((6, cfg.inter_channels, Tx * 4)).astype(np.float32)
    for b in range(6):
        n = int(lengths[b])
        one = eng.run(ids[b:b + 1, :n], [n], [0.667, 1.0, 0.8], forced_durations=forced[b:b + 1, :n], noise_w=nw[b:b + 1, :, :n],
                      noise_z=nz[b:b + 1, :, : n * 4], want_pcm16=True)
        L = int(one["lengths"][0])
        assert L == int(out["lengths"][b])
        assert np.array_equal(one["audio"][0, :L], out["audio"][b, :L]), b
        assert np.array_equal(one["pcm"][0, :L], out["pcm"][b, :L]), b
    eng.close()
