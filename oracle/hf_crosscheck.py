#!/usr/bin/env python3
"""Pin the oracle against an independent implementation of the same graph.

TEST INFRASTRUCTURE.  The reference's own arithmetic (onnxruntime + generator.onnx)
is not available in this environment.  HuggingFace ``transformers``
ships ``VitsModel`` — an independent re-implementation of upstream VITS inference
(text encoder, stochastic duration predictor with the spline flows, length regulator,
residual-coupling flow, HiFi-GAN with ResBlock1).  This script

  1. builds a ``VitsModel`` whose hyper-parameters equal a ``VitsConfig`` with
     ``resblock="1"``, copies one set of seeded synthetic weights into both,
  2. runs both with zero noise scales on identical phoneme ids,
  3. asserts agreement and writes small fixtures to ``tests/golden/`` so the
     ``-m "not gpu"`` suite can re-check the oracle without ``transformers``.

Run from the repo root:  ``python oracle/hf_crosscheck.py``  (needs ``transformers``; CPU only).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mimic3_amd.config import VitsConfig  # noqa: E402
from mimic3_amd import weights as W  # noqa: E402
from oracle.vits_oracle import VitsOracle  # noqa: E402


def build_hf(cfg: VitsConfig, weights):
    from transformers import VitsConfig as HFConfig, VitsModel

    hf_cfg = HFConfig(
        vocab_size=cfg.num_symbols,
        hidden_size=cfg.hidden_channels,
        num_hidden_layers=cfg.n_layers,
        num_attention_heads=cfg.n_heads,
        window_size=cfg.window_size,
        use_bias=True,
        ffn_dim=cfg.filter_channels,
        layerdrop=0.0,
        ffn_kernel_size=cfg.kernel_size,
        flow_size=cfg.inter_channels,
        spectrogram_bins=16,
        hidden_act="relu",
        use_stochastic_duration_prediction=True,
        num_speakers=cfg.n_speakers,
        speaker_embedding_size=cfg.gin_channels if cfg.is_multispeaker else 0,
        upsample_initial_channel=cfg.upsample_initial_channel,
        upsample_rates=list(cfg.upsample_rates),
        upsample_kernel_sizes=list(cfg.upsample_kernel_sizes),
        resblock_kernel_sizes=list(cfg.resblock_kernel_sizes),
        resblock_dilation_sizes=[list(d) for d in cfg.resblock_dilation_sizes],
        leaky_relu_slope=0.1,
        depth_separable_channels=2,
        depth_separable_num_layers=cfg.dp_dds_layers,
        duration_predictor_flow_bins=cfg.dp_num_bins,
        duration_predictor_tail_bound=cfg.dp_tail_bound,
        duration_predictor_kernel_size=cfg.dp_kernel_size,
        duration_predictor_num_flows=cfg.dp_n_flows,
        prior_encoder_num_flows=cfg.flow_n_flows,
        prior_encoder_num_wavenet_layers=cfg.flow_wn_layers,
        posterior_encoder_num_wavenet_layers=2,
        wavenet_kernel_size=cfg.flow_wn_kernel,
        wavenet_dilation_rate=cfg.flow_wn_dilation_rate,
        noise_scale=0.0,
        noise_scale_duration=0.0,
        speaking_rate=1.0,
        layer_norm_eps=1e-5,
        pad_token_id=None,
    )
    torch.manual_seed(0)
    model = VitsModel(hf_cfg).eval()
    sd = model.state_dict()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    new = {}

    def put(key, arr):
        assert key in sd, key
        assert tuple(sd[key].shape) == tuple(arr.shape), (key, sd[key].shape, arr.shape)
        new[key] = t(arr)

    def put_conv(hf, ours, wn=False):
        w = weights[ours + ".weight"]
        if wn and (hf + ".parametrizations.weight.original0") in sd:
            # weight = g * v / ||v||  with the norm over all dims but 0
            g = np.sqrt((w.reshape(w.shape[0], -1) ** 2).sum(1)).reshape(-1, 1, 1)
            put(hf + ".parametrizations.weight.original0", g.astype(np.float32))
            put(hf + ".parametrizations.weight.original1", w)
        elif wn and (hf + ".weight_g") in sd:
            g = np.sqrt((w.reshape(w.shape[0], -1) ** 2).sum(1)).reshape(-1, 1, 1)
            put(hf + ".weight_g", g.astype(np.float32))
            put(hf + ".weight_v", w)
        else:
            put(hf + ".weight", w)
        if ours + ".bias" in weights:
            put(hf + ".bias", weights[ours + ".bias"])

    def put_ln(hf, ours):
        put(hf + ".weight", weights[ours + ".gamma"])
        put(hf + ".bias", weights[ours + ".beta"])

    def put_dds(hf, ours):
        for i in range(cfg.dp_dds_layers):
            put_conv(f"{hf}.convs_dilated.{i}", f"{ours}.convs_sep.{i}")
            put_conv(f"{hf}.convs_pointwise.{i}", f"{ours}.convs_1x1.{i}")
            put_ln(f"{hf}.norms_1.{i}", f"{ours}.norms_1.{i}")
            put_ln(f"{hf}.norms_2.{i}", f"{ours}.norms_2.{i}")

    put("text_encoder.embed_tokens.weight", weights["enc_p.emb.weight"])
    for i in range(cfg.n_layers):
        a = f"text_encoder.encoder.layers.{i}"
        o = f"enc_p.encoder.attn_layers.{i}"
        for hfn, on in (("q_proj", "conv_q"), ("k_proj", "conv_k"), ("v_proj", "conv_v"), ("out_proj", "conv_o")):
            put(f"{a}.attention.{hfn}.weight", weights[f"{o}.{on}.weight"][:, :, 0])
            put(f"{a}.attention.{hfn}.bias", weights[f"{o}.{on}.bias"])
        put(f"{a}.attention.emb_rel_k", weights[f"{o}.emb_rel_k"])
        put(f"{a}.attention.emb_rel_v", weights[f"{o}.emb_rel_v"])
        put_ln(f"{a}.layer_norm", f"enc_p.encoder.norm_layers_1.{i}")
        put_conv(f"{a}.feed_forward.conv_1", f"enc_p.encoder.ffn_layers.{i}.conv_1")
        put_conv(f"{a}.feed_forward.conv_2", f"enc_p.encoder.ffn_layers.{i}.conv_2")
        put_ln(f"{a}.final_layer_norm", f"enc_p.encoder.norm_layers_2.{i}")
    put_conv("text_encoder.project", "enc_p.proj")

    put_conv("duration_predictor.conv_pre", "dp.pre")
    put_conv("duration_predictor.conv_proj", "dp.proj")
    put_dds("duration_predictor.conv_dds", "dp.convs")
    if cfg.is_multispeaker:
        put_conv("duration_predictor.cond", "dp.cond")
    put("duration_predictor.flows.0.translate", weights["dp.flows.0.m"])
    put("duration_predictor.flows.0.log_scale", weights["dp.flows.0.logs"])
    for j in range(1, cfg.dp_n_flows):  # HF flows[1+j] <-> upstream flows[1+2j]
        hf = f"duration_predictor.flows.{1 + j}"
        o = f"dp.flows.{1 + 2 * j}"
        put_conv(hf + ".conv_pre", o + ".pre")
        put_dds(hf + ".conv_dds", o + ".convs")
        put_conv(hf + ".conv_proj", o + ".proj")

    for j in range(cfg.flow_n_flows):
        hf = f"flow.flows.{j}"
        o = f"flow.flows.{2 * j}"
        put_conv(hf + ".conv_pre", o + ".pre")
        for l in range(cfg.flow_wn_layers):
            put_conv(f"{hf}.wavenet.in_layers.{l}", f"{o}.enc.in_layers.{l}", wn=True)
            put_conv(f"{hf}.wavenet.res_skip_layers.{l}", f"{o}.enc.res_skip_layers.{l}", wn=True)
        if cfg.is_multispeaker:
            put_conv(f"{hf}.wavenet.cond_layer", f"{o}.enc.cond_layer", wn=True)
        put_conv(hf + ".conv_post", o + ".post")

    put_conv("decoder.conv_pre", "dec.conv_pre")
    nk = len(cfg.resblock_kernel_sizes)
    for i in range(len(cfg.upsample_rates)):
        put_conv(f"decoder.upsampler.{i}", f"dec.ups.{i}")
        for j in range(nk):
            n = i * nk + j
            for m in range(len(cfg.resblock_dilation_sizes[j])):
                put_conv(f"decoder.resblocks.{n}.convs1.{m}", f"dec.resblocks.{n}.convs1.{m}")
                put_conv(f"decoder.resblocks.{n}.convs2.{m}", f"dec.resblocks.{n}.convs2.{m}")
    put_conv("decoder.conv_post", "dec.conv_post")
    if cfg.is_multispeaker:
        put_conv("decoder.cond", "dec.cond")
        put("embed_speaker.weight", weights["emb_g.weight"])

    missing, unexpected = model.load_state_dict(new, strict=False)
    assert not unexpected, unexpected
    # everything we did not set must be training-only or the unused first ConvFlow
    for k in missing:
        assert k.startswith("posterior_encoder.") or k.startswith("duration_predictor.post_") \
            or k.startswith("duration_predictor.flows.1."), k
    return model


def hf_resblock1_config(base: VitsConfig) -> VitsConfig:
    """HF's HifiGanResidualBlock is ResBlock1 with a fixed second dilation of 1."""
    c = VitsConfig(**{**base.__dict__})
    c.resblock = "1"
    if base.resblock == "2":
        c.resblock_dilation_sizes = tuple((1, 3) for _ in base.resblock_kernel_sizes)
    return c


def compare(cfg: VitsConfig, seed: int, ids: np.ndarray, lengths: np.ndarray, sid=None, tag=""):
    weights = W.synthetic_weights(cfg, seed=seed, frames_per_id=3.0)
    oracle = VitsOracle(cfg, weights)
    res = oracle.infer(ids, lengths, [0.0, 1.0, 0.0], sid=None if sid is None else np.full(len(lengths), sid),
                       batch_semantics="upstream")
    model = build_hf(cfg, weights)
    am = (np.arange(ids.shape[1])[None, :] < lengths[:, None]).astype(np.int64)
    with torch.no_grad():
        out = model(torch.from_numpy(ids), attention_mask=torch.from_numpy(am), speaker_id=sid)
    hf_wave = out.waveform.numpy()
    hf_len = out.sequence_lengths.numpy()
    hf_spec = out.spectrogram.numpy()
    assert np.array_equal(hf_len, res["audio_lengths"]), (hf_len, res["audio_lengths"])
    worst = 0.0
    for b in range(ids.shape[0]):
        L = int(hf_len[b])
        a, r = res["audio"][b, 0, :L], hf_wave[b, :L]
        rel = float(np.sqrt(np.mean((a - r) ** 2)) / max(1e-12, np.sqrt(np.mean(r ** 2))))
        worst = max(worst, rel)
        Ty = L // cfg.upsample_factor
        zs = float(np.abs(res["z"][b, :, :Ty] - hf_spec[b, :, :Ty]).max())
        print(f"[{tag}] utt {b}: L={L} rel_rms(audio)={rel:.3e} max|dz|={zs:.3e}")
        assert rel < 2e-5, rel
        assert zs < 1e-4, zs
    return weights, res, hf_wave, hf_len, hf_spec, worst


def main():
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    rng = np.random.default_rng(7)

    # (1) tiny graph, ragged batch of 3, single speaker: fixture carries weights + HF outputs
    cfg = hf_resblock1_config(VitsConfig.tiny())
    ids = rng.integers(1, cfg.num_symbols, size=(3, 12)).astype(np.int64)
    lengths = np.array([12, 7, 9], dtype=np.int64)
    for b in range(3):
        ids[b, lengths[b]:] = 0
    weights, res, hf_wave, hf_len, hf_spec, _ = compare(cfg, 11, ids, lengths, tag="tiny")
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "hf_tiny_resblock1.npz"),
        config_json=np.array(cfg.to_json()), ids=ids, lengths=lengths,
        hf_waveform=hf_wave, hf_lengths=hf_len, hf_spectrogram=hf_spec,
        **{"w:" + k: v for k, v in weights.items()},
    )

    # (2) tiny graph, multi-speaker, B=1
    cfg_ms = hf_resblock1_config(VitsConfig.tiny(n_speakers=5))
    ids1 = rng.integers(1, cfg_ms.num_symbols, size=(1, 10)).astype(np.int64)
    len1 = np.array([10], dtype=np.int64)
    weights, res, hf_wave, hf_len, hf_spec, _ = compare(cfg_ms, 12, ids1, len1, sid=3, tag="tiny-ms")
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "hf_tiny_resblock1_multispeaker.npz"),
        config_json=np.array(cfg_ms.to_json()), ids=ids1, lengths=len1, sid=np.array([3]),
        hf_waveform=hf_wave, hf_lengths=hf_len, hf_spectrogram=hf_spec,
        **{"w:" + k: v for k, v in weights.items()},
    )

    # (3) full "_low" shapes (resblock 1 variant), B=1, 48 ids: weights are re-derivable from the
    # seed, so the fixture stores only a decimated waveform + lengths.
    cfg_full = hf_resblock1_config(VitsConfig.apope_low())
    ids2 = rng.integers(1, cfg_full.num_symbols, size=(1, 48)).astype(np.int64)
    len2 = np.array([48], dtype=np.int64)
    weights, res, hf_wave, hf_len, hf_spec, worst = compare(cfg_full, 13, ids2, len2, tag="low-rb1")
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "hf_low_resblock1_decimated.npz"),
        config_json=np.array(cfg_full.to_json()), ids=ids2, lengths=len2, seed=np.array(13),
        frames_per_id=np.array(3.0), hf_waveform_dec16=hf_wave[:, ::16], hf_lengths=hf_len,
        weight_checksum=np.array(float(sum(float(np.abs(v).sum()) for v in weights.values()))),
    )
    print("HF cross-check passed; worst rel RMS on full-size graph:", worst)


if __name__ == "__main__":
    main()
