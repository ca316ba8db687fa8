"""A/B tests on the real MI355X between the default kernels and the paths they replaced, through the LAB build of the library
(-DMI355_LAB: the product build reads no kernel-choice switch).  Same C ABI, same sources; the environment variables below are
read when an engine is created."""
import numpy as np
import pytest

from mimic3_amd import weights as W
from mimic3_amd._native import Engine
from mimic3_amd.config import VitsConfig
from tests.util import REL_RMS_TOL, rel_rms

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("voice", ["apope_low", "vctk_low"])
def test_dds_stack_equals_one_launch_per_piece(lab_lib, voice, monkeypatch):
    """k_dds_stack (pre + DDS layers + proj (+ spline) in one launch, x in a 64-column LDS window) vs one launch per piece on the
    device: durations equal, h and logw to f32 rounding on every valid column (the two sets of kernels are compiled separately and
    hipcc contracts their multiply-adds differently; on the CPU model, tests/test_emu_engine.py, they agree bit for bit), audio to
    the parity tolerance; ragged rows that end inside a workgroup's columns, inside a halo, and a one-phoneme row."""
    cfg = VitsConfig.apope_low() if voice == "apope_low" else VitsConfig.vctk_low()
    w = W.synthetic_weights(cfg, seed=31, frames_per_id=1.0)
    blob = W.pack(cfg, w)
    Tx = 150
    rng = np.random.default_rng(3)
    ids = rng.integers(1, cfg.num_symbols, (5, Tx))
    lengths = [Tx, 97, 130, 1, 33]
    sid = np.array([5, 0, 108, 17, 3]) if cfg.is_multispeaker else None
    res = {}
    monkeypatch.setenv("MI355VITS_NO_ENC_GEMM", "1")  # the pieces' 1x1 convs on the f32 matrix cores, as in the stack
    for tag in ("stack", "pieces"):
        if tag == "pieces":
            monkeypatch.setenv("MI355VITS_NO_DDS_STACK", "1")
        eng = Engine(blob, library=lab_lib, device=0)
        eng.profile_enable(True)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], sid, debug_taps=True, seed=11)
        labels = set(eng.profile_report())
        assert ("dp.stack" in labels) == (tag == "stack") and ("dds.layer" in labels) == (tag == "pieces"), labels
        res[tag] = eng.tap("dp.h"), eng.tap("logw"), eng.tap("w_ceil"), out["lengths"].copy(), out["audio"].copy()
        eng.close()
    for bi, L in enumerate(lengths):
        h_s, h_p = res["stack"][0][bi, :, :L], res["pieces"][0][bi, :, :L]
        assert np.abs(h_s - h_p).max() <= 2e-5 * max(1.0, np.abs(h_p).max()), (bi, np.abs(h_s - h_p).max())
        assert np.abs(res["stack"][1][bi, :, :L] - res["pieces"][1][bi, :, :L]).max() < 1e-4
    assert np.array_equal(res["stack"][2], res["pieces"][2]) and np.array_equal(res["stack"][3], res["pieces"][3])
    for bi in range(len(lengths)):
        n = int(res["stack"][3][bi])
        assert rel_rms(res["stack"][4][bi, :n], res["pieces"][4][bi, :n]) < REL_RMS_TOL


def test_encoder_slice_kernel_equals_general_conv_kernels(lab_lib, monkeypatch):
    """k_enc_b3 (q/k/v, o, FFN and the duration predictor's pointwise convs: one 192-channel slice per workgroup, the FFN's second
    conv as four slices added up by the LayerNorm launch) vs the general conv kernels (MI355VITS_NO_ENC_GEMM=1) on the device:
    encoder output, prior statistics and h to f32 rounding, durations and lengths equal, audio to the parity tolerance."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=32, frames_per_id=1.0)
    blob = W.pack(cfg, w)
    Tx = 150
    ids = np.random.default_rng(4).integers(1, cfg.num_symbols, (4, Tx))
    lengths = [Tx, 65, 1, 129]
    res = {}
    for tag in ("slice", "general"):
        if tag == "general":
            monkeypatch.setenv("MI355VITS_NO_ENC_GEMM", "1")
        eng = Engine(blob, library=lab_lib, device=0)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], debug_taps=True, seed=12)
        res[tag] = eng.tap("x"), eng.tap("stats"), eng.tap("dp.h"), eng.tap("w_ceil"), out["lengths"].copy(), out["audio"].copy()
        eng.close()
    for bi, L in enumerate(lengths):
        for k in range(3):
            a, b = res["slice"][k][bi, :, :L], res["general"][k][bi, :, :L]
            assert np.abs(a - b).max() <= 5e-5 * max(1.0, np.abs(b).max()), (bi, k, np.abs(a - b).max())
    assert np.array_equal(res["slice"][3], res["general"][3]) and np.array_equal(res["slice"][4], res["general"][4])
    for bi in range(len(lengths)):
        n = int(res["slice"][4][bi])
        assert rel_rms(res["slice"][5][bi, :n], res["general"][5][bi, :n]) < REL_RMS_TOL


def test_o_proj_layernorm_fused_equals_two_launches(lab_lib, monkeypatch):
    """k_enc_o_ln (o-proj + residual + LayerNorm in one launch, in place) vs conv + LayerNorm (MI355VITS_NO_ENC_O_LN=1) on the device:
    encoder output to f32 rounding, durations and lengths equal, audio to the parity tolerance; rows ending inside a tile."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=33, frames_per_id=1.0)
    blob = W.pack(cfg, w)
    Tx = 150
    ids = np.random.default_rng(5).integers(1, cfg.num_symbols, (4, Tx))
    lengths = [Tx, 65, 1, 129]
    res = {}
    for tag in ("fused", "two"):
        if tag == "two":
            monkeypatch.setenv("MI355VITS_NO_ENC_O_LN", "1")
        eng = Engine(blob, library=lab_lib, device=0)
        eng.profile_enable(True)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], debug_taps=True, seed=13)
        labels = set(eng.profile_report())
        assert ("enc.o_ln" in labels) == (tag == "fused"), labels
        res[tag] = eng.tap("x"), eng.tap("w_ceil"), out["lengths"].copy(), out["audio"].copy()
        eng.close()
    for bi, L in enumerate(lengths):
        a, b = res["fused"][0][bi, :, :L], res["two"][0][bi, :, :L]
        assert np.abs(a - b).max() <= 5e-5 * max(1.0, np.abs(b).max()), (bi, np.abs(a - b).max())
    assert np.array_equal(res["fused"][1], res["two"][1]) and np.array_equal(res["fused"][2], res["two"][2])
    for bi in range(len(lengths)):
        n = int(res["fused"][2][bi])
        assert rel_rms(res["fused"][3][bi, :n], res["two"][3][bi, :n]) < REL_RMS_TOL


def test_flow_pointwise_convs_slice_kernel_equals_general(lab_lib, monkeypatch):
    """flow.pre / flow.post at frame resolution on k_enc_b3 (96- and 192-channel slices) vs the general conv kernels
    (MI355VITS_NO_FLOW_GEMM=1) on the device: z to f32 rounding, lengths equal, audio to the parity tolerance."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=34, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    Tx = 90
    ids = np.random.default_rng(6).integers(1, cfg.num_symbols, (3, Tx))
    lengths = [Tx, 41, 77]
    res = {}
    for tag in ("slice", "general"):
        if tag == "general":
            monkeypatch.setenv("MI355VITS_NO_FLOW_GEMM", "1")
        eng = Engine(blob, library=lab_lib, device=0)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], debug_taps=True, seed=14)
        res[tag] = eng.tap("z"), out["lengths"].copy(), out["audio"].copy()
        eng.close()
    assert np.array_equal(res["slice"][1], res["general"][1])
    for bi in range(3):
        n = int(res["slice"][1][bi])
        fr = n // cfg.hop_length
        a, b = res["slice"][0][bi, :, :fr], res["general"][0][bi, :, :fr]
        assert np.abs(a - b).max() <= 5e-5 * max(1.0, np.abs(b).max()), (bi, np.abs(a - b).max())
        assert rel_rms(res["slice"][2][bi, :n], res["general"][2][bi, :n]) < REL_RMS_TOL


def test_mrf_row_sweeps_equal_block_kernel_bitwise_on_the_device(lab_lib, monkeypatch):
    """The row-sweep MRF kernel k_mrf_s (64 channels: one pass per resblock, conv-specialised waves, fragments in registers, LDS
    rings) vs k_mrf_p (MI355VITS_MRF_SWEEP_SEG=0) on the MI355X at full-size shapes: both MRF stage taps and the waveform BIT FOR
    BIT, so the launcher may pick by grid size.  The product's default and a forced short segment (1,632 columns: several items per
    CU); ragged rows ending inside a segment, a one-phoneme row.  (The 32-channel stage stays on k_mrf_p: its single-pass sweep,
    k_mrf_s1, tied it twice in round 4 and was deleted in round 5.)"""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234)
    blob = W.pack(cfg, w)
    B, Tx = 8, 128
    ids = np.random.default_rng(8).integers(1, 50, (B, Tx))
    lengths = np.array([Tx, Tx, 97, Tx, 64, Tx, 1, 127])
    forced = np.full((B, Tx), 6, np.int32)
    res = {}
    for tag, env in (("default", None), ("block", "0"), ("short_segments", "1632")):
        if env is None:
            monkeypatch.delenv("MI355VITS_MRF_SWEEP_SEG", raising=False)
        else:
            monkeypatch.setenv("MI355VITS_MRF_SWEEP_SEG", env)
        eng = Engine(blob, library=lab_lib, device=0)
        eng.profile_enable(True)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], forced_durations=forced, seed=3, debug_taps=True)
        labels = set(eng.profile_report())
        assert ("dec.mrf_s.s1" in labels) == (tag != "block") and "dec.mrf_s.s2" not in labels, (tag, labels)
        res[tag] = eng.tap("dec.mrf.1"), eng.tap("dec.mrf.2"), out["audio"].copy()
        eng.close()
    for tag in ("default", "short_segments"):
        for k in range(3):
            assert np.array_equal(res[tag][k], res["block"][k]), (tag, k)


def test_resblock_conv_128_channels_on_the_device(lab_lib, monkeypatch):
    """k_rb_conv (the 128-channel MRF stage conv by conv, all input channels resident in LDS, kernels_rbc.cpp) and k_ups_pl / k_ups64 (the three
    upsamplers in the same form) on the MI355X at the full-size shapes: the stage-0 tap and the waveform against the kernels it replaces (MI355VITS_NO_RBC=1: k_mrf_fused + the staged
    conv — another order of summation, so within tolerance on each row's own columns), and its 128- and 32-column work items BIT FOR
    BIT (the launcher picks by grid size; a row's bits must not depend on what it is batched with).  Ragged rows ending inside an
    item, a one-phoneme row."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234)
    blob = W.pack(cfg, w)
    B, Tx = 8, 128
    ids = np.random.default_rng(8).integers(1, 50, (B, Tx))
    lengths = np.array([Tx, Tx, 97, Tx, 64, Tx, 1, 127])
    forced = np.full((B, Tx), 6, np.int32)
    res = {}
    # "wide" = 128-column items on the producer-wave form (k_rb_conv_pw, the default), run TWICE ("wide_again": a race between the producer
    # waves' LDS stores and the matrix waves' reads would show as a difference between two runs — the CPU model cannot see one); "wide_pw0" =
    # the staging inside the matrix waves' streams (k_rb_conv), "wide_pw2" = weight fragments two steps ahead instead of three
    for tag, env in (("default", {}), ("wide", {"MI355VITS_RBC_WIDE": "1"}), ("wide_again", {"MI355VITS_RBC_WIDE": "1"}),
                     ("wide_pw0", {"MI355VITS_RBC_WIDE": "1", "MI355VITS_RBC_PW": "0"}), ("wide_pw2", {"MI355VITS_RBC_WIDE": "1", "MI355VITS_RBC_PW": "2"}),
                     ("narrow", {"MI355VITS_RBC_WIDE": "0"}), ("old", {"MI355VITS_NO_RBC": "1"}),
                     ("wide_o0", {"MI355VITS_RBC_WIDE": "1", "MI355VITS_RBC_ITEM_ORDER": "0"})):  # items w, w + W, ... instead of XCD-major
        for k in ("MI355VITS_RBC_WIDE", "MI355VITS_NO_RBC", "MI355VITS_RBC_PW", "MI355VITS_RBC_ITEM_ORDER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = Engine(blob, library=lab_lib, device=0)
        eng.profile_enable(True)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], forced_durations=forced, seed=3, debug_taps=True)
        labels = set(eng.profile_report())
        assert ("dec.mrf_fused.s0" in labels) == (tag == "old"), (tag, labels)
        res[tag] = eng.tap("dec.mrf.0"), out["audio"].copy(), out["lengths"].copy(), eng.tap("dec.ups.1"), eng.tap("dec.ups.2"), eng.tap("dec.ups.0")
        eng.close()
    for tag in ("default", "narrow", "wide_again", "wide_pw0", "wide_pw2", "wide_o0"):
        for k in (0, 1, 3, 4, 5):
            assert np.array_equal(res[tag][k], res["wide"][k]), (tag, k)
    for bi in range(B):
        n = int(res["old"][2][bi])
        for k in (0, 3, 4, 5):  # the stage-0 MRF output and the three upsamplers (k_ups_pl, k_ups64)
            hop = res["old"][1].shape[1] // res["old"][k].shape[2]
            assert rel_rms(res["wide"][k][bi, :, : n // hop], res["old"][k][bi, :, : n // hop]) < 2e-6, (bi, k)
        assert rel_rms(res["wide"][1][bi, :n], res["old"][1][bi, :n]) < REL_RMS_TOL, bi


def test_round5_memory_bound_kernels_on_the_device(lab_lib, monkeypatch):
    """Round 5's two memory-side rewrites against the kernels they replace, on the MI355X at the full-size shapes, BIT FOR BIT:
    k_conv_post_tanh_dpp (one 16-byte load per lane, channel and tile, the taps' neighbours through DPP wave shifts — `wave_shr:1`
    exists on the device only: the CPU model uses its shuffle) vs the round-1 kernel (MI355VITS_CONV_POST_V1=1), and k_ups64's
    one 16-byte store per tile vs two 8-byte stores (MI355VITS_UPS64_ST8=1).  Waveform, lengths, int16 and the 64 -> 32 upsampler's
    tap; ragged rows (ending inside a tile / an item), a one-phoneme row; each variant also against itself on a second engine."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234)
    blob = W.pack(cfg, w)
    B, Tx = 8, 128
    ids = np.random.default_rng(8).integers(1, 50, (B, Tx))
    lengths = np.array([Tx, Tx, 97, Tx, 64, Tx, 1, 127])
    forced = np.full((B, Tx), 6, np.int32)
    res = {}
    for tag, env in (("new", {}), ("new_again", {}), ("post_v1", {"MI355VITS_CONV_POST_V1": "1"}), ("st8", {"MI355VITS_UPS64_ST8": "1"}),
                     ("both_old", {"MI355VITS_CONV_POST_V1": "1", "MI355VITS_UPS64_ST8": "1"})):
        for k in ("MI355VITS_CONV_POST_V1", "MI355VITS_UPS64_ST8"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = Engine(blob, library=lab_lib, device=0)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], forced_durations=forced, seed=3, debug_taps=True, want_pcm16=True)
        res[tag] = out["audio"].copy(), out["lengths"].copy(), out["pcm"].copy(), eng.tap("dec.ups.2")
        eng.close()
    for tag in ("new_again", "post_v1", "st8", "both_old"):
        for k in range(4):
            assert np.array_equal(res[tag][k], res["new"][k]), (tag, k)


def test_wavenet_layer_weight_ring_depths_agree_bitwise_on_the_device(lab_lib, monkeypatch):
    """k_wn_layer_b3 with its weight fragments three groups ahead (a ring of four buffers: the default since round 5, for the boxes
    whose L2 loses the layer's fragments to the activation stream) against one group ahead (MI355VITS_WN_RING=2) at the bench
    shape's tile form: z and the waveform BIT FOR BIT (the same products in the same order per accumulator)."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=1234)
    blob = W.pack(cfg, w)
    B, Tx = 8, 128
    ids = np.random.default_rng(9).integers(1, 50, (B, Tx))
    lengths = np.array([Tx, 100, Tx, 3, Tx, 77, Tx, 128])
    forced = np.full((B, Tx), 6, np.int32)
    res = {}
    monkeypatch.setenv("MI355VITS_WN_B3_NT", "3")
    for tag, ring in (("ring4", None), ("ring2", "2"), ("ring4_again", None), ("epi0", "e0"), ("epi1", "e1"), ("epi2", "e2"), ("tw", "tw"), ("nt4", "nt4"), ("nt4r2", "nt4r2")):
        monkeypatch.setenv("MI355VITS_WN_B3_NT", "4" if (ring or "").startswith("nt4") else "3")  # (nt4: the 128-column tiles large grids run since round 6)
        monkeypatch.delenv("MI355VITS_WN_RING", raising=False)
        monkeypatch.delenv("MI355VITS_WN_EPI", raising=False)
        monkeypatch.delenv("MI355VITS_WN_TW", raising=False)
        if ring == "tw":  # the two-workgroups-per-CU form (64-column tiles)
            monkeypatch.setenv("MI355VITS_WN_TW", "1")
        elif ring is not None and ring.startswith("e"):  # the epilogue forms (old values one / three tiles ahead / + issued before the gate)
            monkeypatch.setenv("MI355VITS_WN_EPI", ring[1:])
        elif ring == "nt4r2":  # 128-column tiles with the fragments one group ahead (B double-buffered) instead of three (B single-buffered)
            monkeypatch.setenv("MI355VITS_WN_RING", "2")
        elif ring is not None and ring != "nt4":
            monkeypatch.setenv("MI355VITS_WN_RING", ring)
        eng = Engine(blob, library=lab_lib, device=0)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], forced_durations=forced, seed=3, debug_taps=True)
        res[tag] = eng.tap("z"), out["audio"].copy(), out["lengths"].copy()
        eng.close()
    for tag in ("ring2", "ring4_again", "epi0", "epi1", "epi2", "tw", "nt4", "nt4r2"):
        for k in range(3):
            assert np.array_equal(res[tag][k], res["ring4"][k]), (tag, k)


def test_encoder_row_block_loop_is_bitwise_the_one_block_form(lab_lib, monkeypatch):
    """Round 6: on large grids k_enc_b3 walks three 64-row output blocks per workgroup over ONE staged 192-channel slice (the
    launcher's choice; MI355VITS_ENC_ROWLOOP forces it in the lab build) instead of staging and splitting the slice once per row
    block.  An output element's products and their order are the same: encoder output, prior statistics, duration-predictor state,
    durations, z and the waveform BIT FOR BIT against the one-block form, ragged rows included (the flow's pointwise convs at frame
    resolution run the same kernel)."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=33, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    Tx = 140
    ids = np.random.default_rng(5).integers(1, cfg.num_symbols, (6, Tx))
    lengths = [Tx, 65, 1, 129, 128, 77]
    res = {}
    monkeypatch.setenv("MI355VITS_ENC_WIDE", "0")  # (the 64 x 64 form on both sides; the 128-column form has its own test below)
    for tag, env in (("one", "1"), ("loop", "3")):
        monkeypatch.setenv("MI355VITS_ENC_ROWLOOP", env)
        eng = Engine(blob, library=lab_lib, device=0)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], debug_taps=True, seed=13)
        res[tag] = [eng.tap(k) for k in ("x", "stats", "dp.h", "w_ceil", "z_p", "z")] + [out["lengths"].copy(), out["audio"].copy()]
        eng.close()
    for k, (a, b) in enumerate(zip(res["one"], res["loop"])):
        assert np.array_equal(a, b), k


def test_encoder_128_column_form_is_bitwise_the_64_column_form(lab_lib, monkeypatch):
    """Round 6: on grids that give every CU a workgroup the encoder's dense convs (and the flow's pointwise convs at frame resolution)
    run k_enc_b3w — 128 columns x all of the conv's 32-row tiles per workgroup, a row tile's weight fragments streamed by ONE wave and
    used for four column tiles, buffer-addressed epilogue — instead of 64 x 64 tiles (MI355VITS_ENC_WIDE forces either in the lab
    build).  b3_chunk / b3_chunk_lean walk k-groups and taps in the same order per accumulator and the epilogue is operation for
    operation the same: every text-side tap, the durations, z and the waveform BIT FOR BIT, ragged rows included."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=34, frames_per_id=2.0)
    blob = W.pack(cfg, w)
    Tx = 150
    ids = np.random.default_rng(6).integers(1, cfg.num_symbols, (6, Tx))
    lengths = [Tx, 65, 1, 129, 128, 77]
    res = {}
    for tag, env, six8 in (("narrow", "0", None), ("wide", "1", None), ("wide6", "1", "0")):  # (wide6: six-row-tile blocks on six waves, the round-6a form)
        monkeypatch.setenv("MI355VITS_ENC_WIDE", env)
        if six8 is None:
            monkeypatch.delenv("MI355VITS_ENC_SIX8", raising=False)
        else:
            monkeypatch.setenv("MI355VITS_ENC_SIX8", six8)
        eng = Engine(blob, library=lab_lib, device=0)
        eng.profile_enable(True)
        out = eng.run(ids, lengths, [0.667, 1.0, 0.8], debug_taps=True, seed=14)
        res[tag] = [eng.tap(k) for k in ("x", "stats", "dp.h", "w_ceil", "z_p", "z")] + [out["lengths"].copy(), out["audio"].copy()]
        eng.close()
    for tag in ("wide", "wide6"):
        for k, (a, b) in enumerate(zip(res["narrow"], res[tag])):
            assert np.array_equal(a, b), (tag, k)


def test_attention_high_occupancy_form_is_bitwise_the_prefetched_form(lab_lib, monkeypatch):
    """Round 6: on grids of >= 4 workgroups per CU the rel-pos attention runs its 76-register form (six workgroups per CU, operands
    fetched trip by trip) instead of the one that prefetches every operand up front (256 registers, built for batch-1 latency).  The
    MFMA sequence per output is the same: encoder output, prior statistics, durations and the waveform BIT FOR BIT, ragged rows,
    T <= 128 and T <= 256."""
    cfg = VitsConfig.apope_low()
    w = W.synthetic_weights(cfg, seed=35, frames_per_id=1.0)
    blob = W.pack(cfg, w)
    for Tx, lengths in ((120, [120, 65, 1, 119]), (200, [200, 129, 130, 7])):
        ids = np.random.default_rng(7).integers(1, cfg.num_symbols, (4, Tx))
        res = {}
        for tag in ("prefetch", "trips"):
            if tag == "trips":
                monkeypatch.setenv("MI355VITS_ATTN_NO_PREFETCH", "1")
            else:
                monkeypatch.delenv("MI355VITS_ATTN_NO_PREFETCH", raising=False)
            eng = Engine(blob, library=lab_lib, device=0)
            out = eng.run(ids, lengths, [0.667, 1.0, 0.8], debug_taps=True, seed=15)
            res[tag] = [eng.tap(k) for k in ("x", "stats", "w_ceil")] + [out["lengths"].copy(), out["audio"].copy()]
            eng.close()
        for k, (a, b) in enumerate(zip(res["prefetch"], res["trips"])):
            assert np.array_equal(a, b), (Tx, k)
