"""Parity tests proper: the HIP path on a real MI355X, through the C ABI, against the oracle on identical
seeded inputs, against the committed golden vectors, and through size-independent properties at the
BASELINE.json sizes.  Tolerances are the ones stated in tests/util.py (fp32: rel. RMS <= 1e-4,
max abs <= 1e-3, durations exactly equal, int16 bit-exact on the engine's own float output)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mimic3_amd import weights as W
from mimic3_amd._native import Engine
from mimic3_amd.config import VitsConfig
from oracle.vits_oracle import VitsOracle, audio_float_to_int16
from tests.util import REL_RMS_TOL, check_parity, rel_rms

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_native_library_is_the_hip_build(gpu_lib):
    assert "gfx950" in gpu_lib.version()


def test_mfma_fragment_layout_on_device(gpu_hooks):
    assert gpu_hooks.test_mfma_layout() < 1e-3


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("case", [(2, 16, 32, 70, 3, 1), (1, 32, 32, 1000, 7, 12), (1, 6, 70, 129, 5, 2),
                                  (2, 64, 29, 33, 1, 1), (1, 192, 384, 500, 5, 1), (1, 768, 192, 130, 3, 1),
                                  (3, 128, 128, 2000, 7, 3)])
def test_conv1d_kernels(gpu_hooks, impl, case):
    B, Cin, Cout, T, K, dil = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, Cout, T)).astype(np.float32)
    in_len = np.array([T] + [max(1, T - 3)] * (B - 1), np.int32)
    y = gpu_hooks.test_conv1d(x, w, bias, res, dilation=dil, impl=impl, in_len=in_len, out_len=in_len, in_slope=0.1,
                            out_scale=0.5)
    tm = (torch.arange(T)[None, :] < torch.from_numpy(in_len.astype(np.int64))[:, None]).double()[:, None, :]
    xt = F.leaky_relu(torch.from_numpy(x).double() * tm, 0.1)
    ref = F.conv1d(xt, torch.from_numpy(w).double(), torch.from_numpy(bias).double(), dilation=dil, padding=(K * dil - dil) // 2)
    ref = ((ref + torch.from_numpy(res).double()) * 0.5 * tm).numpy()
    assert np.abs(y - ref).max() < 5e-5, np.abs(y - ref).max()


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("case", [(1, 256, 128, 100, 16, 8), (2, 128, 64, 333, 16, 8), (1, 64, 32, 1000, 8, 4), (1, 6, 3, 5, 4, 2)])
def test_conv_transpose1d(gpu_hooks, case, impl):
    B, Cin, Cout, Tin, K, s = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, Tin)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(2 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    y = gpu_hooks.test_conv_transpose1d(x, w, b, s, in_slope=0.1, impl=impl)
    ref = F.conv_transpose1d(F.leaky_relu(torch.from_numpy(x).double(), 0.1), torch.from_numpy(w).double(),
                             torch.from_numpy(b).double(), stride=s, padding=(K - s) // 2).numpy()
    assert np.abs(y - ref).max() < 5e-5


@pytest.mark.parametrize("case", [(1, 256, 128, 100, 16, 8), (2, 128, 64, 333, 16, 8), (3, 64, 32, 20000, 8, 4), (32, 64, 32, 1500, 8, 4)])
def test_conv_transpose1d_split_bf16_persistent(gpu_hooks, case):
    """impl 2: polyphase upsampler on the split-bf16 persistent producer / consumer kernel (k_conv1d_b3_pc): fewer tiles
    than CUs, and several tiles per workgroup (the last two cases: 315 and 768 tiles on 256 CUs), vs fp64."""
    B, Cin, Cout, Tin, K, s = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, Tin)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(2 * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    y = gpu_hooks.test_conv_transpose1d(x, w, b, s, in_slope=0.1, impl=2)
    ref = F.conv_transpose1d(F.leaky_relu(torch.from_numpy(x).double(), 0.1), torch.from_numpy(w).double(),
                             torch.from_numpy(b).double(), stride=s, padding=(K - s) // 2).numpy()
    assert np.abs(y - ref).max() < 5e-6


@pytest.mark.parametrize("cfgname", ["tiny", "tiny_ms", "tiny_rb1"])
def test_tiny_graphs_match_oracle(gpu_lib, cfgname):
    cfg = {"tiny": VitsConfig.tiny(), "tiny_ms": VitsConfig.tiny(n_speakers=4), "tiny_rb1": VitsConfig.tiny(resblock="1")}[cfgname]
    check_parity(gpu_lib, cfg, B=3, Tx=11, seed=1)
    check_parity(gpu_lib, cfg, B=2, Tx=10, seed=4, noise=True)


def _engine_in(math, cfg, w):
    eng = Engine(W.pack(cfg, w))
    eng.set_math(math)
    return eng


MATH_MODES = ["bf16x3", "f32"]  # the default path and the pure f32-MFMA path: both at the same tolerances


@pytest.mark.parametrize("math", MATH_MODES)
def test_apope_low_b1_matches_oracle(gpu_lib, math):
    """configs[1] shape class: en_UK/apope_low graph, B = 1, natural durations, deterministic scales."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=121, frames_per_id=3.0)
    eng = _engine_in(math, cfg, w)
    check_parity(gpu_lib, cfg, B=1, Tx=40, seed=21, weights=w, engine=eng)
    eng.close()


def test_apope_low_b1_injected_noise_matches_oracle(gpu_lib):
    check_parity(gpu_lib, VitsConfig.apope_low(), B=1, Tx=24, seed=22, frames_per_id=3.0, noise=True)


@pytest.mark.parametrize("math", MATH_MODES)
def test_vctk_low_ragged_batch_matches_oracle(gpu_lib, math):
    """configs[2] shape class: multi-speaker graph (109 speakers, gin 512), ragged batch."""
    cfg = VitsConfig.vctk_low()
    w = W.synthetic_weights(cfg, seed=123, frames_per_id=2.5)
    eng = _engine_in(math, cfg, w)
    check_parity(gpu_lib, cfg, B=3, Tx=20, seed=23, weights=w, engine=eng)
    eng.close()


def test_committed_golden_vector(gpu_lib):
    """Engine vs the oracle output frozen in tests/golden (made by oracle/make_golden.py)."""
    g = np.load(os.path.join(GOLDEN, "oracle_apope_low_b1.npz"))
    cfg = VitsConfig.from_json(str(g["config_json"]))
    w = W.synthetic_weights(cfg, seed=int(g["seed"]), frames_per_id=float(g["frames_per_id"]))
    assert abs(float(sum(float(np.abs(v).sum()) for v in w.values())) - float(g["weight_checksum"])) < 1e-3 * float(g["weight_checksum"])
    eng = Engine(W.pack(cfg, w))
    out = eng.run(g["ids"], g["lengths"], [0, 1, 0], want_pcm16=True)
    assert np.array_equal(out["lengths"], g["audio_lengths"])
    L = int(out["lengths"][0])
    assert rel_rms(out["audio"][0, :L], g["audio"][0, :L]) < REL_RMS_TOL
    d = np.abs(out["pcm"][0, :L].astype(np.int32) - g["pcm"][0, :L].astype(np.int32))
    assert (d > 0).mean() <= 0.10 and d.max() <= 16
    eng.close()


@pytest.mark.parametrize("math", MATH_MODES)
def test_full_size_properties_b8_forced(gpu_lib, math):
    """BASELINE.json sizes (Tx = 128, forced 6 frames/id -> 768 frames, 196,608 samples per row) through
    properties that do not need the oracle: batch invariance (bitwise), determinism, per-row pcm16 ==
    audio_float_to_int16(float row), Philox noise independent of the batch split."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234)
    eng = _engine_in(math, cfg, w)
    B, Tx = 8, 128
    ids = np.stack([np.random.default_rng(1234 + b).integers(1, 50, Tx) for b in range(B)]).astype(np.int64)
    lengths = np.full(B, Tx, np.int64)
    forced = np.full((B, Tx), 6, np.int32)
    sc = [0.667, 1.0, 0.8]
    full = eng.run(ids, lengths, sc, forced_durations=forced, seed=7, utterance_base=0, want_pcm16=True)
    assert list(full["lengths"]) == [768 * 256] * B
    assert np.isfinite(full["audio"]).all() and np.abs(full["audio"]).max() <= 1.0
    again = eng.run(ids, lengths, sc, forced_durations=forced, seed=7, utterance_base=0, want_pcm16=True)
    assert np.array_equal(full["audio"], again["audio"])
    part = eng.run(ids[5:7], lengths[5:7], sc, forced_durations=forced[5:7], seed=7, utterance_base=5, want_pcm16=True)
    assert np.array_equal(part["audio"], full["audio"][5:7])
    for b in range(B):
        assert np.array_equal(full["pcm"][b], audio_float_to_int16(full["audio"][b]))
        assert np.abs(full["pcm"][b]).max() >= 32766  # peak * (32767 / peak) may round just below 32767 in fp32
    eng.close()


def test_golden_shape_utterance_matches_oracle(gpu_lib):
    """Shape of the reference's golden utterance (tests/apope_sample_*.wav: 991 frames = 253,696 samples) — the
    bench's batch-1 latency workload (bench.py "latency_b1"), stochastic scales with injected noise, vs the oracle:
    durations, encoder / prior / flow taps, every decoder stage, waveform, int16 (tests/samples_match.py criterion)."""
    cfg = VitsConfig.apope_low()
    Tx = 180
    ids = np.random.default_rng(99).integers(1, 50, (1, Tx)).astype(np.int64)
    forced = np.full((1, Tx), 5, np.int32)
    forced[0, :91] = 6  # 91*6 + 89*5 = 991 frames
    out, _ = check_parity(gpu_lib, cfg, ids=ids, forced=forced, noise=True, seed=99,
                          weights=W.synthetic_weights(cfg, seed=1234))
    assert int(out["lengths"][0]) == 253696
    assert np.abs(out["pcm"]).max() >= 32766


def _bench_batch(B, Tx, base=0):
    from bench import make_batch  # the very ids bench.py times

    return make_batch(B, Tx, base)


@pytest.mark.parametrize("math", MATH_MODES)
def test_bench_workload_apope_low_b32_matches_oracle(gpu_lib, math):
    """THE benchmarked configuration (bench.py default: en_UK/apope_low, 32 utterances x 128 ids, forced 6 frames/id,
    scales [0.667, 1, 0.8]) against the oracle with both Gaussian draws injected: every row's waveform, int16, the
    taps, and the decoder stages of three rows."""
    cfg = VitsConfig.apope_low()
    B, Tx = 32, 128
    ids, lengths = _bench_batch(B, Tx)
    forced = np.full((B, Tx), 6, np.int32)
    w = W.synthetic_weights(cfg, seed=1234)
    eng = _engine_in(math, cfg, w)
    out, _ = check_parity(gpu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=5, weights=w,
                          stage_rows=(0, 13, 31), engine=eng)
    eng.close()
    assert list(out["lengths"]) == [768 * 256] * B


def test_bench_workload_vctk_low_b32_matches_oracle(gpu_lib):
    """BASELINE.json configs[2]: en_US/vctk_low (109 speakers, gin 512), 32 x 128 ids, sid = b mod 109 — the
    `bench.py` "vctk_low_b32" line's workload — vs the oracle, injected noise."""
    cfg = VitsConfig.vctk_low()
    B, Tx = 32, 128
    ids, lengths = _bench_batch(B, Tx)
    forced = np.full((B, Tx), 6, np.int32)
    sid = (np.arange(B) % cfg.n_speakers).astype(np.int64)
    check_parity(gpu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=6, sid=sid,
                 weights=W.synthetic_weights(cfg, seed=1234), stage_rows=(1, 30))


def test_default_modelconfig_graph_matches_oracle(gpu_lib):
    """The reference's ModelConfig defaults (mimic3_tts/config.py:112-139): ResBlock1, kernels 3/7/11 with dilations
    (1,3,5), four upsample stages 8-8-2-2 from 512 channels — the 'high quality' voice family."""
    cfg = VitsConfig(num_symbols=60, resblock="1", resblock_kernel_sizes=(3, 7, 11),
                     resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)), upsample_rates=(8, 8, 2, 2),
                     upsample_initial_channel=512, upsample_kernel_sizes=(16, 16, 4, 4))
    check_parity(gpu_lib, cfg, B=2, Tx=14, seed=41, frames_per_id=2.0)


@pytest.mark.parametrize("Tx", [1, 33, 300, 600])
def test_sequence_length_extremes(gpu_lib, Tx):
    """One phoneme; just over one MFMA tile; 16 key tiles in registers (T <= 512); beyond that the VALU attention."""
    check_parity(gpu_lib, VitsConfig.tiny(), B=1, Tx=Tx, seed=50 + Tx, frames_per_id=1.1)


@pytest.mark.parametrize("Tx", [1, 31, 64, 65, 129, 257, 600])
def test_sequence_length_extremes_at_the_real_hidden_width(gpu_lib, Tx):
    """The 192-channel text-side kernels (k_enc_b3 slices, k_enc_o_ln, k_dds_stack windows, attention with prefetched operands up to
    256 phonemes and the trip-by-trip / VALU forms beyond) at one phoneme, one column short of / exactly / one column past a 64-column
    tile, across the attention kernel's key-tile variants; ragged second row; noise on (the duration predictor's flows run)."""
    cfg = VitsConfig.tiny_h192()
    rng = np.random.default_rng(70 + Tx)
    ids = rng.integers(1, cfg.num_symbols, (2, Tx))
    lengths = np.array([Tx, max(1, Tx - 7)])
    check_parity(gpu_lib, cfg, ids=ids, lengths=lengths, noise=True, seed=70 + Tx, frames_per_id=1.1)


def test_length_scale_and_rate(gpu_lib):
    check_parity(gpu_lib, VitsConfig.tiny_wide(), B=2, Tx=12, seed=61, scales=(0.0, 1.37, 0.0))


@pytest.mark.parametrize("math", MATH_MODES)
def test_long_form_full_size_utterance(gpu_lib, math):
    """600 phoneme ids at the real voice shapes (more than the 512-key register budget of the MFMA attention, so the
    fallback attention runs at head dimension 96) -> 1800 frames = 20.9 s of audio through every fused kernel."""
    cfg = VitsConfig.apope_low()
    Tx = 600
    forced = np.full((1, Tx), 3, np.int32)
    w = W.synthetic_weights(cfg, seed=177)
    eng = _engine_in(math, cfg, w)
    out, ora = check_parity(gpu_lib, cfg, B=1, Tx=Tx, seed=77, forced=forced, taps=False, ragged=False, weights=w, engine=eng)
    eng.close()
    assert int(out["lengths"][0]) == Tx * 3 * 256


def test_wavenet_layer_geometries_give_identical_bits(gpu_lib):
    """The fused WaveNet-layer kernel picks 6 waves x 2 tiles for small grids (one utterance) and 4 waves x 3 tiles for
    grids that fill the chip; a row's bits must not depend on which one ran: the same utterance alone and as row 0 of a batch
    of 48 (the product library has no switch to force either — the grid decides)."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234)
    eng = Engine(W.pack(cfg, w))
    eng.set_math("f32")  # the geometries belong to the f32 fused layer kernel
    ids = np.random.default_rng(3).integers(1, 50, (1, 48)).astype(np.int64)
    forced = np.full((1, 48), 4, np.int32)
    alone = eng.run(ids, [48], [0.667, 1.0, 0.8], forced_durations=forced, seed=9)["audio"]
    many = eng.run(np.repeat(ids, 48, axis=0), [48] * 48, [0.667, 1.0, 0.8], forced_durations=np.repeat(forced, 48, axis=0), seed=9)["audio"]
    assert np.array_equal(alone[0], many[0])
    eng.close()


def test_plain_c_client_on_the_device(gpu_lib, tmp_path):
    """The C ABI driven from plain C against libmi355vits.so on the MI355X (full-size multi-speaker voice)."""
    from tests.util import run_c_client

    out = run_c_client(gpu_lib, tmp_path, cfg=VitsConfig.vctk_low(), seed=3)
    assert "gfx950" in out and "speakers 109" in out


@pytest.mark.parametrize("voice", ["apope_low", "vctk_low"])
def test_bf16x3_math_bench_workload_matches_oracle(gpu_lib, voice):
    """MATH_BF16X3 (f32 operands split exactly into 3 x bf16, six partial products on the bf16 matrix cores, f32
    accumulate) at the benchmarked configurations, against the oracle at the UNCHANGED f32 tolerances (rel. RMS <= 1e-4,
    durations exact, int16 criterion, every decoder stage)."""
    cfg = VitsConfig.apope_low() if voice == "apope_low" else VitsConfig.vctk_low()
    B, Tx = 32, 128
    ids, lengths = _bench_batch(B, Tx)
    forced = np.full((B, Tx), 6, np.int32)
    sid = (np.arange(B) % cfg.n_speakers).astype(np.int64) if cfg.is_multispeaker else None
    w = W.synthetic_weights(cfg, seed=1234)
    eng = Engine(W.pack(cfg, w))
    eng.set_math("bf16x3")
    out, ora = check_parity(gpu_lib, cfg, ids=ids, lengths=lengths, forced=forced, noise=True, seed=5, sid=sid, weights=w,
                            stage_rows=(0, 17), engine=eng)
    # error level next to the f32-MFMA path's on the same inputs: the split must not cost accuracy
    eng.set_math("f32")
    out32, _ = check_parity(gpu_lib, cfg, ids=ids[:2], lengths=lengths[:2], forced=forced[:2], noise=True, seed=5,
                            sid=None if sid is None else sid[:2], weights=w, engine=eng)
    e3 = rel_rms(out["audio"][0], ora["audio"][0, 0])
    print(f"\n{voice}: rel RMS vs oracle  bf16x3 {e3:.3e}   stages {out['stage_errors']}   f32 stages {out32['stage_errors']}")
    assert max(out["stage_errors"].values()) < 3 * max(out32["stage_errors"].values()) + 2e-6
    eng.close()


def test_bf16x3_golden_shape_and_batch_invariance(gpu_lib):
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234)
    eng = Engine(W.pack(cfg, w))
    eng.set_math("bf16x3")
    Tx = 180
    ids = np.random.default_rng(99).integers(1, 50, (1, Tx)).astype(np.int64)
    forced = np.full((1, Tx), 5, np.int32)
    forced[0, :91] = 6
    out, _ = check_parity(gpu_lib, cfg, ids=ids, forced=forced, noise=True, seed=99, weights=w, engine=eng)
    assert int(out["lengths"][0]) == 253696
    # batched == unbatched, bitwise, in this mode too
    B = 3
    idb = np.random.default_rng(5).integers(1, 50, (B, 40)).astype(np.int64)
    lens = np.array([40, 23, 31])
    fb = np.full((B, 40), 4, np.int32)
    full = eng.run(idb, lens, [0.667, 1.0, 0.8], forced_durations=fb, seed=7)
    one = eng.run(idb[1:2], lens[1:2], [0.667, 1.0, 0.8], forced_durations=fb[1:2], seed=7, utterance_base=1)
    L = int(one["lengths"][0])
    assert np.array_equal(full["audio"][1, :L], one["audio"][0, :L])
    eng.close()


@pytest.mark.parametrize("case", [(2, 192, 384, 1000, 5, 1), (1, 128, 128, 3000, 7, 3), (3, 96, 192, 700, 1, 1), (1, 64, 29, 2000, 3, 9),
                                  (2, 256, 32, 520, 7, 1)])
def test_split_bf16_staged_conv_kernel_vs_fp64(gpu_hooks, case):
    """k_conv1d_b3 (impl 2: f32 operands split 3 x bf16 while staging, six bf16-MFMA products, f32 accumulate) against
    an fp64 conv, next to the f32-MFMA kernel (impl 1) on the same data: at least as accurate."""
    B, Cin, Cout, T, K, dil = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, Cout, T)).astype(np.float32)
    in_len = np.array([T] + [max(1, T - 37)] * (B - 1), np.int32)
    tm = (torch.arange(T)[None, :] < torch.from_numpy(in_len.astype(np.int64))[:, None]).double()[:, None, :]
    xt = F.leaky_relu(torch.from_numpy(x).double() * tm, 0.1)
    ref = F.conv1d(xt, torch.from_numpy(w).double(), torch.from_numpy(bias).double(), dilation=dil, padding=(K * dil - dil) // 2)
    ref = ((ref + torch.from_numpy(res).double()) * 0.5 * tm).numpy()
    err = {}
    for impl in (1, 2):
        y = gpu_hooks.test_conv1d(x, w, bias, res, dilation=dil, impl=impl, in_len=in_len, out_len=in_len, in_slope=0.1, out_scale=0.5)
        err[impl] = float(np.sqrt(np.mean((y - ref) ** 2)))
    print(f"\nconv {case}: rms error vs fp64  f32-MFMA {err[1]:.3e}  split-bf16 {err[2]:.3e}")
    assert err[2] < 2e-6 and err[2] <= 1.25 * err[1] + 2e-8, err


RBC_SHAPES = [(3, 1), (3, 2), (5, 2), (5, 6), (7, 3), (7, 12)]


@pytest.mark.parametrize("kd", RBC_SHAPES)
@pytest.mark.parametrize("grid", ["wide", "narrow"])
def test_resident_input_resblock_conv_vs_fp64(gpu_hooks, kd, grid):
    """k_rb_conv_pw / k_rb_conv (impl 4: one conv of the 128-channel ResBlock2 stage with every input channel resident in LDS, MATH_BF16X3)
    against an fp64 conv, next to the f32-MFMA kernel (impl 1) on the same data — all six (taps, dilation) instantiations, the
    128-column items of large grids (producer-wave form: waves 8 .. 11 stage, waves 0 .. 7 only multiply) and the 32-column items
    of small ones, ragged rows (one ending inside an item, one a column short), writing and accumulating (`y +=`: the stage's second
    and third resblock).  At least as accurate as f32 arithmetic: a kernel that dropped one of its six partial products (l x h:
    ~1e-5 of the output) fails this by a factor of 40."""
    K, dil = kd
    B, T = (9, 3800) if grid == "wide" else (2, 700)  # 9 x 30 = 270 items of 128 columns >= 256 CUs | 44 items of 32 columns
    C = 128
    rng = np.random.default_rng(1000 * K + dil + B)
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    w = (rng.standard_normal((C, C, K)) / np.sqrt(C * K)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32)
    res = rng.standard_normal((B, C, T)).astype(np.float32)
    y0 = rng.standard_normal((B, C, T)).astype(np.float32)
    in_len = np.array([T, T - 61] + [T - 1] * (B - 2), np.int32)
    tm = (torch.arange(T)[None, :] < torch.from_numpy(in_len.astype(np.int64))[:, None]).double()[:, None, :]
    xt = F.leaky_relu(torch.from_numpy(x).double() * tm, 0.1)
    conv = F.conv1d(xt, torch.from_numpy(w).double(), torch.from_numpy(bias).double(), dilation=dil, padding=(K * dil - dil) // 2)
    ref = ((conv + torch.from_numpy(res).double()) * (1.0 / 3.0)).numpy()
    for acc in (False, True):
        want = ref + (y0.astype(np.float64) if acc else 0.0)
        err = {}
        for impl in (1, 4):
            y = gpu_hooks.test_conv1d(x, w, bias, res, dilation=dil, impl=impl, in_len=in_len, in_slope=0.1, out_scale=1.0 / 3.0,
                                    accumulate_into=y0 if acc else None)
            # over each row's own columns: items at or past a row's length are not computed (ragged batches, round 6)
            err[impl] = float(np.sqrt(np.mean(((y - want) * tm.numpy()) ** 2)))
        print(f"\nrb_conv k{K} d{dil} {grid} acc={acc}: rms error vs fp64  f32-MFMA {err[1]:.3e}  resident-input split {err[4]:.3e}")
        assert err[4] < 1e-6 and err[4] <= 1.25 * err[1] + 2e-8, (acc, err)


@pytest.mark.parametrize("case", [(256, 128, 8, 16), (128, 64, 8, 16), (64, 32, 4, 8)])
@pytest.mark.parametrize("grid", ["wide", "narrow"])
def test_resident_input_upsamplers_vs_fp64(gpu_hooks, case, grid):
    """k_ups_pl / k_ups64 (conv-transpose impl 3: the three upsamplers of the "_low" decoder as two-tap polyphase convs with every
    input channel resident, MATH_BF16X3; 16-byte phase-interleaved stores) against an fp64 ConvTranspose1d, next to the f32-MFMA
    polyphase kernel (impl 1) on the same data: wide and narrow work items, first / last output positions included."""
    Cin, Cout, stride, K = case
    per_item = 64 if Cin == 256 else 127 if Cin == 64 else 128
    B, Tin = (5, 52 * per_item + 17) if grid == "wide" else (2, 300)
    rng = np.random.default_rng(Cin + B)
    x = rng.standard_normal((B, Cin, Tin)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(Cin * K / stride)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    ref = F.conv_transpose1d(F.leaky_relu(torch.from_numpy(x).double(), 0.1), torch.from_numpy(w).double(), torch.from_numpy(bias).double(),
                             stride=stride, padding=(K - stride) // 2).numpy()
    err = {}
    for impl in (1, 3):
        y = gpu_hooks.test_conv_transpose1d(x, w, bias, stride, in_slope=0.1, impl=impl)
        assert y.shape == ref.shape
        err[impl] = float(np.sqrt(np.mean((y - ref) ** 2)))
    print(f"\nupsampler {case} {grid}: rms error vs fp64  f32-MFMA {err[1]:.3e}  resident-input split {err[3]:.3e}")
    assert err[3] < 1e-6 and err[3] <= 1.25 * err[1] + 2e-8, err


def test_f16x2_mode_bench_workload_matches_oracle(gpu_lib):
    """MATH_F16X2 (experimental: fused MRF stages on two-term fp16 operands) at the benchmarked shape, every decoder stage
    tapped, default tolerances; and batched == unbatched bitwise."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234, frames_per_id=6.0)
    eng = Engine(W.pack(cfg, w))
    eng.set_math("f16x2")
    B, Tx = 32, 128
    ids = np.stack([np.random.default_rng(1234 + b).integers(1, 50, Tx) for b in range(B)]).astype(np.int64)
    forced = np.full((B, Tx), 6, np.int32)
    out, _ = check_parity(gpu_lib, cfg, ids=ids, forced=forced, noise=True, seed=1234, weights=w, engine=eng, stage_rows=(0, 31))
    one = eng.run(ids[7:8], np.array([Tx]), [0.667, 1.0, 0.8], forced_durations=forced[7:8], seed=5, utterance_base=7)
    full = eng.run(ids, np.full(B, Tx), [0.667, 1.0, 0.8], forced_durations=forced, seed=5)
    assert np.array_equal(full["audio"][7], one["audio"][0])
    eng.close()


def test_bf16_weights_mode_at_its_own_tolerance(gpu_lib):
    """BASELINE.json configs[4] first slice: MATH_BF16W — bf16-rounded weights (leading split term), exact f32 activations,
    f32 accumulate on v_mfma_f32_32x32x16_bf16 — on the full-size apope_low graph, bench-shaped rows.  Reduced precision:
    its OWN tolerance (rel. RMS <= 2e-2 vs the f32 oracle, durations still exactly equal: the text side — encoder, duration
    predictor — runs the exact three-term split in this mode too, Engine::tmath), and demonstrably not f32-grade (so it can
    never be mistaken for the default)."""
    cfg = VitsConfig.apope_low()
    B, Tx = 4, 128
    ids, lengths = _bench_batch(B, Tx)
    forced = np.full((B, Tx), 6, np.int32)
    w = W.synthetic_weights(cfg, seed=1234)
    rng = np.random.default_rng(11)
    nw = rng.standard_normal((B, 2, Tx)).astype(np.float32)
    nz = rng.standard_normal((B, cfg.inter_channels, Tx * 6)).astype(np.float32)
    sc = (0.667, 1.0, 0.8)
    ora = VitsOracle(cfg, w).infer(ids, lengths, sc, noise_w=nw, noise_z=nz, forced_durations=forced, stage_rows=())
    eng = _engine_in("bf16w", cfg, w)
    out = eng.run(ids, lengths, sc, noise_w=nw, noise_z=nz, forced_durations=forced, want_pcm16=True, debug_taps=True)
    assert np.array_equal(out["lengths"], ora["audio_lengths"]) and np.array_equal(eng.tap("w_ceil"), ora["w_ceil"])
    errs = [rel_rms(out["audio"][b, : int(out["lengths"][b])], ora["audio"][b, 0, : int(out["lengths"][b])]) for b in range(B)]
    print(f"\nbf16-weights mode: rel RMS vs f32 oracle {max(errs):.3e}")
    assert 2e-5 < max(errs) < 2e-2, errs
    for b in range(B):
        L = int(out["lengths"][b])
        assert np.array_equal(out["pcm"][b, :L], audio_float_to_int16(out["audio"][b, :L]))
    eng.close()


def test_bf16_weights_mode_natural_durations_200_sentences(gpu_lib):
    """MATH_BF16W at NATURAL durations (BASELINE configs[4] is long-form text: the lengths are what the duration predictor says):
    208 sentences of 40-160 ids, stochastic duration noise on.  ceil(exp(logw) * length_scale) is discontinuous, so the mode
    keeps the whole text side on the exact three-term split: durations, lengths, encoder output and prior statistics are BITWISE
    those of the default mode for every sentence (0 length mismatches), and the first batch's durations equal the f32 oracle's.
    Reference: mimic3_tts/voice.py:182-189 (scales) -> the frame counts that size every later tensor."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234)
    rng = np.random.default_rng(404)
    B, NB = 16, 13
    sc = (0.667, 1.0, 0.8)
    engs = {m: _engine_in(m, cfg, w) for m in ("bf16x3", "bf16w")}
    mismatches, total = 0, 0
    for nb in range(NB):
        lengths = rng.integers(40, 161, B).astype(np.int64)
        Tx = int(lengths.max())
        ids = rng.integers(1, cfg.num_symbols, (B, Tx)).astype(np.int64)
        for b in range(B):
            ids[b, lengths[b]:] = 0
        nw = rng.standard_normal((B, 2, Tx)).astype(np.float32)
        got = {}
        for m, eng in engs.items():
            out = eng.run(ids, lengths, sc, noise_w=nw, seed=5, utterance_base=nb * B, debug_taps=True)
            got[m] = (out["lengths"].copy(), eng.tap("w_ceil"), eng.tap("x"), eng.tap("stats"))
        mismatches += int((got["bf16x3"][0] != got["bf16w"][0]).sum())
        total += B
        for k in range(1, 4):
            assert np.array_equal(got["bf16x3"][k], got["bf16w"][k]), (nb, k)
        if nb == 0:
            ora = VitsOracle(cfg, w).infer(ids, lengths, (0.0, sc[1], sc[2]), noise_w=nw, stage_rows=())  # durations do not depend on noise_scale
            assert np.array_equal(got["bf16w"][1], ora["w_ceil"]) and np.array_equal(got["bf16w"][0], ora["audio_lengths"])
    print(f"\nbf16w natural durations: {mismatches} length mismatches in {total} sentences")
    assert total >= 200 and mismatches == 0
    for e in engs.values():
        e.close()


ENC_CASES = [(2, 192, 576, 70, 1), (1, 192, 192, 1, 1), (2, 96, 192, 130, 1), (1, 96, 40, 65, 3), (3, 192, 768, 130, 3), (2, 768, 192, 65, 3), (1, 192, 29, 64, 1),
             (1, 384, 100, 33, 3)]


@pytest.mark.parametrize("case", ENC_CASES)
def test_encoder_slice_kernel_vs_fp64(gpu_hooks, case):
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
    y = gpu_hooks.test_conv1d(x, w, bias, res, impl=3, in_len=in_len, out_len=None if split else in_len)
    tm = (torch.arange(T)[None, :] < torch.from_numpy(in_len.astype(np.int64))[:, None]).double()[:, None, :]
    ref = F.conv1d(torch.from_numpy(x).double() * tm, torch.from_numpy(w).double(), None if split else torch.from_numpy(bias).double(),
                   padding=(K - 1) // 2)
    if not split:
        ref = (ref + torch.from_numpy(res).double()) * tm
    assert np.abs(y - ref.numpy()).max() < 5e-5, np.abs(y - ref.numpy()).max()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_f32_grade_modes_are_within_3x_of_pytorch_fp32_against_fp64(mode):
    """The self-calibrating accuracy guard (tests/util.py f32_grade_vs_fp64; round 5): on the same inputs the engine's error against
    the fp64 oracle must stay within 3 x the error PyTorch-CPU fp32 has against it.  On the CPU model a split kernel that loses even its
    SMALLEST partial product (l_w x h_x: an error the 5e-6 tap bound lets through) lands at 5.6 x and fails this
    (tests/test_emu_engine.py::test_a_dropped_partial_product_is_caught); the intact kernels measure 1.5 x (split) / 1.7 x (f32 MFMA) at
    the bench shapes.  apope_low, 4 x 64 ids x 6 frames per id, both Gaussian draws injected."""
    import torch

    from tests.util import f32_grade_vs_fp64

    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234)
    B, Tx = 4, 64
    rng = np.random.default_rng(77)
    ids = rng.integers(1, 50, (B, Tx))
    lengths = np.array([Tx, 40, Tx, 17])
    forced = np.full((B, Tx), 6, np.int32)
    nw = rng.standard_normal((B, 2, Tx)).astype(np.float32)
    nz = rng.standard_normal((B, cfg.inter_channels, Tx * 6)).astype(np.float32)
    scales = (0.667, 1.0, 0.8)
    o64 = VitsOracle(cfg, w, dtype=torch.float64).infer(ids, lengths, scales, noise_w=nw, noise_z=nz, forced_durations=forced)
    o32 = VitsOracle(cfg, w).infer(ids, lengths, scales, noise_w=nw, noise_z=nz, forced_durations=forced)
    eng = Engine(W.pack(cfg, w), device=0)
    eng.set_math(mode)
    out = eng.run(ids, lengths, scales, forced_durations=forced, noise_w=nw, noise_z=nz)
    eng.close()
    ok, e, e32 = f32_grade_vs_fp64(out, o64, o32)
    assert ok, (mode, e, e32)
    assert e < 5e-6, e
