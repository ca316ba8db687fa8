"""CPU oracle for the Mimic 3 hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file, and only as the checker / reported CPU baseline.  The
product path (``mimic3_amd``) never imports it and fails loudly when the HIP
library is missing.

What it restates
----------------
The reference executes ``self.onnx_model.run(None, inputs)[0].squeeze()`` followed by
``audio_float_to_int16`` (``mimic3_tts/voice.py:229-232``).  The arithmetic of ``run``
lives in two un-vendored third-party artefacts: onnxruntime (``requirements.txt:6``,
``onnxruntime>=1.6,<2.0``) and the per-voice ``generator.onnx`` exported by the
``vits_train`` trainer from upstream VITS (``mimic3_tts/const.py:22-24``).  Neither is
present in this environment, so this file restates the published
VITS inference graph in plain PyTorch fp32
(fp64 selectable for error attribution), tensor names = upstream state-dict keys.

Pinning status (see DESIGN.md "Oracle")
---------------------------------------
* ``audio_float_to_int16``: pinned against the reference's own function
  (``mimic3_tts/utils.py:237-244`` imported by path; fixtures in ``tests/golden/``).
* text encoder, SDP (incl. spline inverse), length regulator, flow, HiFi-GAN with
  ResBlock1: pinned against an independent implementation of the same upstream
  graph, HuggingFace ``transformers`` ``VitsModel`` (``oracle/hf_crosscheck.py`` copies
  weights and compares end to end; fixtures in ``tests/golden/``).
* ResBlock2 (what the ``_low`` voices use) and the ``scales``/``sid`` feed: restated from
  upstream VITS from memory; no executable reference is available here ->
  **parity unpinned** for those two details.
* The reference's own golden WAVs (``tests/apope_sample_*.wav``) pin the *whole*
  text->wav pipeline with real weights and cannot be replayed here; their
  statistics calibrate the tolerances.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # upstream modules.LRELU_SLOPE


def _t(x, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(dtype)
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


class VitsOracle:
    """Functional restatement of the VITS inference graph (K1-K12 of SURVEY.md §8a)."""

    def __init__(self, cfg, weights: Dict[str, np.ndarray], dtype=torch.float32):
        cfg.validate()
        self.cfg = cfg
        self.dtype = dtype
        self.w = {k: _t(v, dtype) for k, v in weights.items()}

    # ------------------------------------------------------------------ primitives
    def _conv(self, name, x, dilation=1, padding=0):
        """A.1 Conv1d; weight [C_out, C_in, K] (HF modeling_vits.py:326, 436-437)."""
        return F.conv1d(x, self.w[name + ".weight"], self.w.get(name + ".bias"), dilation=dilation, padding=padding)

    def _ln(self, name, x, eps=1e-5):
        """A.3 channel LayerNorm over dim 1 (HF:1047-1049, 626-627)."""
        mean = x.mean(1, keepdim=True)
        var = ((x - mean) ** 2).mean(1, keepdim=True)
        xn = (x - mean) * torch.rsqrt(var + eps)
        return xn * self.w[name + ".gamma"].view(1, -1, 1) + self.w[name + ".beta"].view(1, -1, 1)

    # ------------------------------------------------------------------ K1-K4 text encoder
    def _attention(self, i, x, x_mask):
        """A.4 relative-position MHA (HF:844-997; upstream attentions.MultiHeadAttention)."""
        cfg = self.cfg
        p = f"enc_p.encoder.attn_layers.{i}"
        B, C, T = x.shape
        nh = cfg.n_heads
        hd = C // nh
        Wn = cfg.window_size
        q = self._conv(p + ".conv_q", x).view(B, nh, hd, T).transpose(2, 3)  # [B,h,T,d]
        k = self._conv(p + ".conv_k", x).view(B, nh, hd, T).transpose(2, 3)
        v = self._conv(p + ".conv_v", x).view(B, nh, hd, T).transpose(2, 3)
        qs = q / math.sqrt(hd)
        scores = torch.matmul(qs, k.transpose(-2, -1))  # [B,h,T,T]
        ek = self.w[p + ".emb_rel_k"][0]  # [2W+1, d]
        ev = self.w[p + ".emb_rel_v"][0]
        idx = torch.arange(T)
        rel = idx[None, :] - idx[:, None]  # j - i
        inwin = rel.abs() <= Wn
        relc = (rel + Wn).clamp(0, 2 * Wn)
        # logits: (q_i/sqrt(d)) . E_k[j-i+W] inside the window, 0 outside (HF:963-985)
        rl = torch.matmul(qs, ek.t())  # [B,h,T,2W+1]
        bias = torch.gather(rl, -1, relc.view(1, 1, T, T).expand(B, nh, T, T))
        scores = scores + bias * inwin.view(1, 1, T, T).to(scores.dtype)
        am = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)  # [B,1,T,T]
        scores = scores.masked_fill(am == 0, -1e4)
        pattn = torch.softmax(scores, dim=-1)
        out = torch.matmul(pattn, v)  # [B,h,T,d]
        # values: sum_{|j-i|<=W} p[i,j] * E_v[j-i+W]  (HF:946-950, 987-997)
        pw = pattn * inwin.view(1, 1, T, T).to(pattn.dtype)
        onehot = F.one_hot(relc, 2 * Wn + 1).to(pattn.dtype)  # [T,T,2W+1]
        relw = torch.einsum("bhij,ijr->bhir", pw, onehot)
        out = out + torch.matmul(relw, ev)
        out = out.transpose(2, 3).reshape(B, C, T)
        return self._conv(p + ".conv_o", out)

    def _ffn(self, i, x, x_mask):
        """A.5 conv-FFN with 'same' padding (HF:1012-1036)."""
        p = f"enc_p.encoder.ffn_layers.{i}"
        k = self.cfg.kernel_size
        pl, pr = (k - 1) // 2, k // 2
        y = self._conv(p + ".conv_1", F.pad(x * x_mask, (pl, pr)))
        y = torch.relu(y)
        y = self._conv(p + ".conv_2", F.pad(y * x_mask, (pl, pr)))
        return y * x_mask

    def text_encoder(self, ids, lengths):
        cfg = self.cfg
        B, T = ids.shape
        x = self.w["enc_p.emb.weight"][ids] * math.sqrt(cfg.hidden_channels)  # K1 (HF:1171)
        x = x.transpose(1, 2)  # [B,H,T]
        x_mask = (torch.arange(T)[None, :] < lengths[:, None]).to(self.dtype).unsqueeze(1)
        x = x * x_mask
        for i in range(cfg.n_layers):  # post-LN (HF:1058-1071)
            y = self._attention(i, x, x_mask)
            x = self._ln(f"enc_p.encoder.norm_layers_1.{i}", x + y)
            y = self._ffn(i, x, x_mask)
            x = self._ln(f"enc_p.encoder.norm_layers_2.{i}", x + y)
        x = x * x_mask
        stats = self._conv("enc_p.proj", x) * x_mask  # K4 (HF:1184-1185)
        m_p, logs_p = torch.split(stats, cfg.inter_channels, dim=1)
        return x, m_p, logs_p, x_mask

    # ------------------------------------------------------------------ K5 stochastic duration predictor
    def _dds(self, prefix, x, x_mask, g=None):
        """A.6 DDSConv (HF:629-643)."""
        cfg = self.cfg
        if g is not None:
            x = x + g
        k = cfg.dp_kernel_size
        for i in range(cfg.dp_dds_layers):
            d = k ** i
            pad = (k * d - d) // 2
            y = F.conv1d(x * x_mask, self.w[f"{prefix}.convs_sep.{i}.weight"], self.w[f"{prefix}.convs_sep.{i}.bias"],
                         dilation=d, padding=pad, groups=x.shape[1])
            y = F.gelu(self._ln(f"{prefix}.norms_1.{i}", y))
            y = self._conv(f"{prefix}.convs_1x1.{i}", y)
            y = F.gelu(self._ln(f"{prefix}.norms_2.{i}", y))
            x = x + y
        return x * x_mask

    def _spline_inverse(self, x1, uw, uh, ud):
        """A.7 rational-quadratic spline, reverse direction, linear tails (HF:93-302)."""
        cfg = self.cfg
        tb = cfg.dp_tail_bound
        nb = cfg.dp_num_bins
        min_w = min_h = min_d = 1e-3
        inside = (x1 >= -tb) & (x1 <= tb)
        const = math.log(math.exp(1 - min_d) - 1)
        ud = F.pad(ud, (1, 1))
        ud[..., 0] = const
        ud[..., -1] = const

        widths = torch.softmax(uw, dim=-1)
        widths = min_w + (1 - min_w * nb) * widths
        cw = F.pad(torch.cumsum(widths, -1), (1, 0))
        cw = 2 * tb * cw - tb
        cw[..., 0] = -tb
        cw[..., -1] = tb
        widths = cw[..., 1:] - cw[..., :-1]
        derivs = min_d + F.softplus(ud)
        heights = torch.softmax(uh, dim=-1)
        heights = min_h + (1 - min_h * nb) * heights
        chh = F.pad(torch.cumsum(heights, -1), (1, 0))
        chh = 2 * tb * chh - tb
        chh[..., 0] = -tb
        chh[..., -1] = tb
        heights = chh[..., 1:] - chh[..., :-1]

        locs = chh.clone()
        locs[..., -1] += 1e-6
        xin = x1.clamp(-tb, tb)  # only evaluated where `inside`; clamp keeps the gather in range
        bin_idx = (torch.sum(xin[..., None] >= locs, dim=-1) - 1).clamp(0, nb - 1)[..., None]
        g = lambda t: t.gather(-1, bin_idx)[..., 0]
        in_cw, in_w = g(cw), g(widths)
        in_ch, in_h = g(chh), g(heights)
        delta = heights / widths
        in_delta = g(delta)
        in_d = g(derivs)
        in_d1 = derivs[..., 1:].gather(-1, bin_idx)[..., 0]
        t1 = in_d + in_d1 - 2 * in_delta
        u = xin - in_ch
        t3 = u * t1
        a = in_h * (in_delta - in_d) + t3
        b = in_h * in_d - t3
        c = -in_delta * u
        disc = b * b - 4 * a * c
        root = (2 * c) / (-b - torch.sqrt(disc))
        out = root * in_w + in_cw
        return torch.where(inside, out, x1)

    def _convflow_reverse(self, idx, z, x_mask, h):
        """A.7 ConvFlow, reverse (HF:658-686)."""
        cfg = self.cfg
        p = f"dp.flows.{idx}"
        nb = cfg.dp_num_bins
        fc = cfg.hidden_channels
        x0, x1 = z[:, :1], z[:, 1:]
        t = self._conv(p + ".pre", x0)
        t = self._dds(p + ".convs", t, x_mask, g=h)
        t = self._conv(p + ".proj", t) * x_mask  # [B, 3nb-1, T]
        B, _, T = x0.shape
        t = t.reshape(B, 1, -1, T).permute(0, 1, 3, 2)  # [B,1,T,3nb-1]
        uw = t[..., :nb] / math.sqrt(fc)
        uh = t[..., nb:2 * nb] / math.sqrt(fc)
        ud = t[..., 2 * nb:]
        x1 = self._spline_inverse(x1, uw, uh, ud)
        return torch.cat([x0, x1], 1) * x_mask

    def duration_predictor(self, x, x_mask, g, noise_w, noise):
        """K5 reverse pass (HF:740-804): returns logw [B,1,T]."""
        cfg = self.cfg
        h = self._conv("dp.pre", x)
        if g is not None:
            h = h + self._conv("dp.cond", g)
        h = self._dds("dp.convs", h, x_mask)
        h = self._conv("dp.proj", h) * x_mask
        z = noise * noise_w  # [B,2,T]
        # reversed upstream list with the "useless" flow removed: Flip,CF_n,...,Flip,CF_2,Flip,EA
        for j in range(cfg.dp_n_flows - 1, 0, -1):
            z = torch.flip(z, [1])
            z = self._convflow_reverse(1 + 2 * j, z, x_mask, h)
        z = torch.flip(z, [1])
        z = (z - self.w["dp.flows.0.m"]) * torch.exp(-self.w["dp.flows.0.logs"]) * x_mask  # EA^-1 (HF:703)
        return z[:, :1]

    # ------------------------------------------------------------------ K8 flow
    def _wn(self, prefix, h, mask, g):
        """A.9 WaveNet stack (HF:347-374)."""
        cfg = self.cfg
        H = cfg.hidden_channels
        k = cfg.flow_wn_kernel
        out = torch.zeros_like(h)
        gc = self._conv(prefix + ".cond_layer", g) if g is not None else None
        for i in range(cfg.flow_wn_layers):
            d = cfg.flow_wn_dilation_rate ** i
            a = self._conv(f"{prefix}.in_layers.{i}", h, dilation=d, padding=(k * d - d) // 2)
            if gc is not None:
                a = a + gc[:, i * 2 * H:(i + 1) * 2 * H]
            u = torch.tanh(a[:, :H]) * torch.sigmoid(a[:, H:])
            rs = self._conv(f"{prefix}.res_skip_layers.{i}", u)
            if i < cfg.flow_wn_layers - 1:
                h = (h + rs[:, :H]) * mask
                out = out + rs[:, H:]
            else:
                out = out + rs
        return out * mask

    def flow_reverse(self, z, y_mask, g):
        """A.9 residual coupling block, reverse (HF:563-597)."""
        cfg = self.cfg
        half = cfg.half_channels
        for j in range(cfg.flow_n_flows - 1, -1, -1):
            z = torch.flip(z, [1])
            p = f"flow.flows.{2 * j}"
            x0, x1 = z[:, :half], z[:, half:]
            h = self._conv(p + ".pre", x0) * y_mask
            h = self._wn(p + ".enc", h, y_mask, g)
            mean = self._conv(p + ".post", h) * y_mask
            x1 = (x1 - mean) * y_mask
            z = torch.cat([x0, x1], 1)
        return z

    # ------------------------------------------------------------------ K9-K12 HiFi-GAN
    def decoder(self, z, g, return_stages=False):
        """A.10 (HF:534-551) with ResBlock1 (HF:455-463) or ResBlock2 (upstream, from memory)."""
        cfg = self.cfg
        stages = {}
        x = self._conv("dec.conv_pre", z, padding=3)
        if g is not None:
            x = x + self._conv("dec.cond", g)
        stages["dec.conv_pre"] = x
        nk = len(cfg.resblock_kernel_sizes)
        for i, (r, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            x = F.leaky_relu(x, LRELU_SLOPE)
            x = F.conv_transpose1d(x, self.w[f"dec.ups.{i}.weight"], self.w[f"dec.ups.{i}.bias"], stride=r,
                                   padding=(k - r) // 2)
            stages[f"dec.ups.{i}"] = x
            xs = None
            for j, (rk, rd) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
                n = i * nk + j
                y = x
                if cfg.resblock == "2":
                    for m, d in enumerate(rd):
                        yt = F.leaky_relu(y, LRELU_SLOPE)
                        yt = self._conv(f"dec.resblocks.{n}.convs.{m}", yt, dilation=d, padding=(rk * d - d) // 2)
                        y = yt + y
                else:
                    for m, d in enumerate(rd):
                        yt = F.leaky_relu(y, LRELU_SLOPE)
                        yt = self._conv(f"dec.resblocks.{n}.convs1.{m}", yt, dilation=d, padding=(rk * d - d) // 2)
                        yt = F.leaky_relu(yt, LRELU_SLOPE)
                        yt = self._conv(f"dec.resblocks.{n}.convs2.{m}", yt, dilation=1, padding=(rk - 1) // 2)
                        y = yt + y
                xs = y if xs is None else xs + y
            x = xs / nk
            stages[f"dec.mrf.{i}"] = x
        x = F.leaky_relu(x)  # default slope 0.01 (HF:548)
        x = self._conv("dec.conv_post", x, padding=3)
        x = torch.tanh(x)
        if return_stages:
            return x, stages
        return x

    # ------------------------------------------------------------------ whole graph
    @torch.no_grad()
    def infer(
        self,
        ids: np.ndarray,
        lengths: np.ndarray,
        scales,
        sid: Optional[np.ndarray] = None,
        noise_w: Optional[np.ndarray] = None,
        noise_z: Optional[np.ndarray] = None,
        forced_durations: Optional[np.ndarray] = None,
        batch_semantics: str = "per_row",
        stage_rows=None,
    ) -> Dict[str, np.ndarray]:
        """Restates ``onnx_model.run(None, {"input","input_lengths","scales"[,"sid"]})``
        (feed built at ``voice.py:180-218``; scales = [noise_scale, length_scale, noise_w]).

        ``noise_w`` [B,2,Tx] / ``noise_z`` [B,C,>=Ty] are the two Gaussian draws of A.12
        (required when the corresponding scale is non-zero; the reference's RNG stream is
        onnxruntime-internal, so parity uses injected noise or zero scales).
        ``forced_durations`` int [B,Tx] overrides ceil(exp(logw)*length_scale) (bench mode F).
        ``batch_semantics``: the reference only ever feeds B = 1 (``voice.py:180``).  Upstream's decoder is
        unmasked, so in a padded batch the tail of a shorter row would see its neighbours' padding;
        ``"per_row"`` (default) decodes every row over its own frames only, i.e. a batch equals B separate
        B = 1 calls — the contract of the drop-in.  ``"upstream"`` keeps the unmasked batch
        decode (used to cross-check against HF ``VitsModel``).
        ``stage_rows``: rows whose per-stage decoder tensors (``dec.conv_pre``, ``dec.ups.i``, ``dec.mrf.i``) are
        kept under ``"stages"`` as ``{row: {name: [C, T_row]}}`` (``per_row`` only; default: all rows, packed into
        padded ``[B, C, T]`` arrays under the stage names as before) — at batch 32 x 768 frames the full set is 2.6 GB.
        Returns every intermediate the parity tests compare.
        """
        cfg = self.cfg
        ids_t = torch.as_tensor(np.asarray(ids), dtype=torch.long)
        len_t = torch.as_tensor(np.asarray(lengths), dtype=torch.long)
        B, Tx = ids_t.shape
        ns, ls, nw = (float(s) for s in np.asarray(scales, dtype=np.float32))
        g = None
        if cfg.is_multispeaker:
            if sid is None:
                raise ValueError("multi-speaker voice needs sid")
            g = self.w["emb_g.weight"][torch.as_tensor(np.asarray(sid), dtype=torch.long)].unsqueeze(-1)  # A.11
        x, m_p, logs_p, x_mask = self.text_encoder(ids_t, len_t)
        if nw != 0.0:
            if noise_w is None:
                raise ValueError("noise_w scale != 0 needs injected noise_w")
            nwt = _t(noise_w, self.dtype)
        else:
            nwt = torch.zeros(B, 2, Tx, dtype=self.dtype)
        logw = self.duration_predictor(x, x_mask, g, torch.tensor(nw, dtype=self.dtype), nwt)
        # K6 length regulator (HF:1350-1371): (exp(logw) * mask) * length_scale, ceil
        w = torch.exp(logw) * x_mask * torch.tensor(ls, dtype=self.dtype)
        w_ceil = torch.ceil(w)
        if forced_durations is not None:
            w_ceil = torch.as_tensor(np.asarray(forced_durations), dtype=self.dtype).view(B, 1, Tx) * x_mask
        y_len = torch.clamp_min(w_ceil.sum([1, 2]), 1).long()
        Ty = int(y_len.max())
        y_mask = (torch.arange(Ty)[None, :] < y_len[:, None]).to(self.dtype).unsqueeze(1)
        cum = torch.cumsum(w_ceil[:, 0], -1)  # [B,Tx]
        tt = torch.arange(Ty, dtype=self.dtype)
        # frame t -> phoneme j(t) = #{j : cum_j <= t}  (A.8)
        j_of_t = (cum[:, None, :] <= tt[None, :, None]).sum(-1).clamp(max=Tx - 1)  # [B,Ty]
        gidx = j_of_t[:, None, :].expand(B, cfg.inter_channels, Ty)
        m_pe = torch.gather(m_p, 2, gidx) * y_mask
        logs_pe = torch.gather(logs_p, 2, gidx) * y_mask
        if ns != 0.0:
            if noise_z is None:
                raise ValueError("noise_scale != 0 needs injected noise_z")
            nz = _t(noise_z, self.dtype)[:, :, :Ty]
        else:
            nz = torch.zeros_like(m_pe)
        z_p = m_pe + nz * torch.exp(logs_pe) * torch.tensor(ns, dtype=self.dtype)  # K7 (HF:1373)
        z = self.flow_reverse(z_p, y_mask, g)
        zm = z * y_mask
        if batch_semantics == "upstream" or B == 1:
            audio, stages = self.decoder(zm, g, return_stages=True)  # [B,1,L]
            if stage_rows is not None and 0 not in stage_rows:
                stages = {}
        elif batch_semantics == "per_row":
            audio = torch.zeros(B, 1, Ty * cfg.upsample_factor, dtype=self.dtype)
            stages = {}
            row_stages = {}
            for b in range(B):
                nb = int(y_len[b])
                ab, sb = self.decoder(zm[b:b + 1, :, :nb], None if g is None else g[b:b + 1], return_stages=True)
                audio[b, :, : ab.shape[-1]] = ab[0]
                if stage_rows is not None:
                    if b in stage_rows:
                        row_stages[b] = {k: v[0].detach().cpu().numpy() for k, v in sb.items()}
                    continue
                for k, v in sb.items():
                    if k not in stages:
                        f = v.shape[-1] // nb
                        stages[k] = torch.zeros(B, v.shape[1], Ty * f, dtype=self.dtype)
                    stages[k][b, :, : v.shape[-1]] = v[0]
        else:
            raise ValueError(batch_semantics)
        out = {
            "x": x, "m_p": m_p, "logs_p": logs_p, "logw": logw, "w_ceil": w_ceil, "y_lengths": y_len,
            "z_p": z_p, "z": z * y_mask, "audio": audio,
        }
        out.update(stages)
        res = {k: v.detach().cpu().numpy() for k, v in out.items()}
        res["audio_lengths"] = (y_len * cfg.upsample_factor).numpy()
        if stage_rows is not None and batch_semantics == "per_row" and B > 1:
            res["stages"] = row_stages
        return res


def audio_float_to_int16(audio: np.ndarray, max_wav_value: float = 32767.0) -> np.ndarray:
    """Restates ``mimic3_tts/utils.py:237-244`` (A2): peak-normalise, clip, truncating cast."""
    audio = np.asarray(audio, dtype=np.float32)
    peak = np.float32(max(0.01, np.max(np.abs(audio)))) if audio.size else np.float32(0.01)
    norm = audio * (np.float32(max_wav_value) / peak)  # float32 divide, float32 multiply
    norm = np.clip(norm, -max_wav_value, max_wav_value)
    return norm.astype("int16")


def run_like_reference(oracle: VitsOracle, feed: Dict[str, np.ndarray], **kw):
    """``onnx_model.run(None, feed)`` -> ``[float32 [B,1,L]]`` exactly as the reference consumes it
    (``voice.py:230``)."""
    r = oracle.infer(feed["input"], feed["input_lengths"], feed["scales"], feed.get("sid"), **kw)
    return [r["audio"].astype(np.float32)]
