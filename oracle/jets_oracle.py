"""fp32 CPU restatement of JETSGenerator.forward (inference branch), B = 1.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function cites the
reference lines it follows (paths relative to /root/reference).  Written as
pure functions over a {key: tensor} state dict; no nn.Module, no batching:
ragged batches are evaluated one utterance at a time, which is the reference's
actual call pattern (inference_am_vocoder_joint.py:115-129) and the parity
semantics the HIP engine must reproduce per utterance.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

ORACLE_TAPS = (
    "tok_emb", "enc_l0", "enc_l1", "enc_l2", "enc_l3", "enc_out", "x_proj", "pitch", "energy",
    "log_dur", "dur", "mel_len", "x_var", "upsampled", "dec_l0", "dec_l1", "dec_l2", "dec_l3",
    "dec_out", "mel", "voc_pre", "voc_up0", "voc_up1", "voc_up2", "voc_up3",
    "voc_mrf0", "voc_mrf1", "voc_mrf2", "voc_mrf3", "wav",
)

LN_EPS = 1e-12  # modules/encoder.py:116


def _t(x):
    if isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(x))


def to_torch_sd(sd) -> Dict[str, torch.Tensor]:
    return {k: _t(v).float() if _t(v).is_floating_point() else _t(v) for k, v in sd.items()}


# --------------------------------------------------------------------------- AM blocks

def sinusoid_table(T: int, d: int) -> torch.Tensor:
    """modules/encoder.py:216-237: pe[t,2i]=sin(t*exp(-2i*ln(1e4)/d)), pe[t,2i+1]=cos(...)."""
    pos = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(T, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def layer_norm(x, sd, prefix):
    """modules/encoder.py:112-127 (eps = 1e-12), over the channel (last) dim."""
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], LN_EPS)


def self_attention(x, sd, prefix, heads):
    """modules/encoder.py:72-109 with an all-valid mask (B = 1)."""
    T, D = x.shape
    dk = D // heads
    q = F.linear(x, sd[prefix + ".linear_q.weight"], sd[prefix + ".linear_q.bias"]).view(T, heads, dk).transpose(0, 1)
    k = F.linear(x, sd[prefix + ".linear_k.weight"], sd[prefix + ".linear_k.bias"]).view(T, heads, dk).transpose(0, 1)
    v = F.linear(x, sd[prefix + ".linear_v.weight"], sd[prefix + ".linear_v.bias"]).view(T, heads, dk).transpose(0, 1)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)       # :108
    attn = torch.softmax(scores, dim=-1)                                  # :94-95
    ctx = torch.matmul(attn, v).transpose(0, 1).contiguous().view(T, D)   # :98-101
    return F.linear(ctx, sd[prefix + ".linear_out.weight"], sd[prefix + ".linear_out.bias"])


def conv_ffn(x, sd, prefix):
    """modules/encoder.py:50-52: Conv1d(k,p=(k-1)/2) -> exact GELU -> Conv1d."""
    w1, w2 = sd[prefix + ".w_1.weight"], sd[prefix + ".w_2.weight"]
    h = F.conv1d(x.t().unsqueeze(0), w1, sd[prefix + ".w_1.bias"], padding=(w1.shape[-1] - 1) // 2)
    h = F.gelu(h)
    h = F.conv1d(h, w2, sd[prefix + ".w_2.bias"], padding=(w2.shape[-1] - 1) // 2)
    return h.squeeze(0).t()


def encoder_stack(x, sd, prefix, n_layers, heads, taps=None, tap_prefix=None):
    """modules/encoder.py:316-324 (embed -> layers -> after_norm);
    layer body :154-200 with normalize_before=True, concat_after=False."""
    T, D = x.shape
    x = x + sd[prefix + ".embed.0.alpha"] * sinusoid_table(T, D)          # :257-261
    for i in range(n_layers):
        p = f"{prefix}.encoders.{i}"
        x = x + self_attention(layer_norm(x, sd, p + ".norm1"), sd, p + ".self_attn", heads)
        x = x + conv_ffn(layer_norm(x, sd, p + ".norm2"), sd, p + ".feed_forward")
        if taps is not None:
            taps[f"{tap_prefix}_l{i}"] = x
    return layer_norm(x, sd, prefix + ".after_norm")


def predictor_trunk(x, sd, prefix, n_layers):
    """modules/variance.py:41-46 / :115-119: n x [Conv1d k3 -> ReLU -> LayerNorm(channels)] -> Linear(C,1)."""
    h = x.t().unsqueeze(0)
    for i in range(n_layers):
        w = sd[f"{prefix}.conv.{i}.0.weight"]
        h = F.conv1d(h, w, sd[f"{prefix}.conv.{i}.0.bias"], padding=(w.shape[-1] - 1) // 2)
        h = torch.relu(h)
        h = layer_norm(h.transpose(1, 2), sd, f"{prefix}.conv.{i}.2").transpose(1, 2)
    out = F.linear(h.squeeze(0).t(), sd[prefix + ".linear.weight"], sd[prefix + ".linear.bias"])
    return out.squeeze(-1)


def duration_from_log(log_d):
    """modules/variance.py:47-51: clamp(round(exp(x) - 1.0), min=0).long() (round half to even)."""
    return torch.clamp(torch.round(log_d.exp() - 1.0), min=0).long()


def gaussian_upsampling(hs, ds, alpha=1.0, delta=0.1):
    """modules/alignment.py:180-211 for B = 1, h_masks=None, all tokens valid."""
    ds = ds * alpha
    if ds.sum() == 0:                                                     # :187-191
        ds = torch.ones_like(ds)
    T_feats = int(torch.sum(ds).int().item())                             # :194-195
    t = torch.arange(0, T_feats).float()
    c = ds.cumsum(dim=-1) - ds / 2                                        # :202
    energy = -1 * delta * (t.unsqueeze(-1) - c.unsqueeze(0)) ** 2         # :204
    p_attn = torch.softmax(energy, dim=1)                                 # :209
    return torch.matmul(p_attn, hs), T_feats                              # :210


def am_forward(sd, ling, speaker, style, content, shapes, alpha=1.0, taps=None, durations=None, duration_scale=1.0):
    """models/prompt_tts_modified/model_open_source.py:102-147 (mel_targets=None branch).

    ling (N,) int64; speaker int; style/content (768,) fp32.  ``durations`` (N,) int64
    overrides the predicted durations (teacher-forced test mode)."""
    H = shapes.hidden
    tok = sd["am.src_word_emb.weight"][ling]                              # :107
    if taps is not None:
        taps["tok_emb"] = tok
    x = encoder_stack(tok, sd, "am.encoder", shapes.enc_layers, shapes.heads, taps, "enc")  # :108
    if taps is not None:
        taps["enc_out"] = x
    N = x.shape[0]
    spk = sd["am.spk_tokenizer.weight"][speaker]                          # :109
    cat = torch.cat([x, spk.expand(N, -1), style.expand(N, -1), content.expand(N, -1)], dim=-1)  # :110
    x = F.linear(cat, sd["am.embed_projection1.weight"], sd["am.embed_projection1.bias"])       # :111
    p_outs = predictor_trunk(x, sd, "am.pitch_predictor", shapes.pitch_layers)                  # :120
    e_outs = predictor_trunk(x, sd, "am.energy_predictor", shapes.energy_layers)                # :121
    log_d = predictor_trunk(x, sd, "am.duration_predictor", shapes.dur_layers)                  # :130
    d_outs = duration_from_log(log_d) if durations is None else durations
    kp = sd["am.pitch_embed.0.weight"].shape[-1]
    p_emb = F.conv1d(p_outs.view(1, 1, -1), sd["am.pitch_embed.0.weight"], sd["am.pitch_embed.0.bias"],
                     padding=(kp - 1) // 2).squeeze(0).t()                                       # :131
    e_emb = F.conv1d(e_outs.view(1, 1, -1), sd["am.energy_embed.0.weight"], sd["am.energy_embed.0.bias"],
                     padding=(kp - 1) // 2).squeeze(0).t()                                       # :132
    if taps is not None:
        taps.update(x_proj=x, pitch=p_outs, energy=e_outs, log_dur=log_d, dur=d_outs)
    x = x + p_emb + e_emb                                                                        # :134
    if taps is not None:
        taps["x_var"] = x
    # :142 -- the inference branch calls ``self.length_regulator(x, d_outs, None, ~src_mask)`` WITHOUT alpha: the ``alpha``
    # argument of JETSGenerator.forward only reaches GaussianUpsampling in the teacher-forced branch (:138), so it has no
    # effect here (pinned by tests/golden/n24_alpha1p3.npz, generated by the reference with alpha = 1.3).
    # ``duration_scale`` is the extension ev_synthesize exposes: GaussianUpsampling.forward's own alpha (alignment.py:183).
    del alpha
    up, T = gaussian_upsampling(x, d_outs, duration_scale)
    if taps is not None:
        taps["upsampled"] = up
        taps["mel_len"] = torch.tensor(T)
    y = encoder_stack(up, sd, "am.decoder", shapes.dec_layers, shapes.heads, taps, "dec")       # :146 (mask=None)
    if taps is not None:
        taps["dec_out"] = y
    mel = F.linear(y, sd["am.to_mel.weight"], sd["am.to_mel.bias"])                              # :147
    if taps is not None:
        taps["mel"] = mel
    return dict(dec_outputs=mel, pitch_predictions=p_outs, energy_predictions=e_outs,
                log_duration_predictions=d_outs, log_dur_raw=log_d, mel_len=T)


# --------------------------------------------------------------------------- vocoder

def fold_weight_norm(sd, prefix):
    """models/hifigan/models.py:10-14 (weight_norm, dim=0): w = g * v / ||v||_2, norm over dims != 0.
    Accepts torch>=2.1 keys (parametrizations.weight.original0/1) and legacy weight_g/weight_v."""
    if prefix + ".parametrizations.weight.original0" in sd:
        g, v = sd[prefix + ".parametrizations.weight.original0"], sd[prefix + ".parametrizations.weight.original1"]
    elif prefix + ".weight_g" in sd:
        g, v = sd[prefix + ".weight_g"], sd[prefix + ".weight_v"]
    else:
        return sd[prefix + ".weight"]
    nrm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * v / nrm


def hifigan_forward(sd, mel_ct, shapes, taps=None, prefix="generator"):
    """models/hifigan/models.py:115-131.  mel_ct: (80, T) fp32 -> wav (256*T,)."""
    x = mel_ct.unsqueeze(0)
    x = F.conv1d(x, fold_weight_norm(sd, prefix + ".conv_pre"), sd[prefix + ".conv_pre.bias"], padding=3)  # :116
    if taps is not None:
        taps["voc_pre"] = x.squeeze(0)
    nk = len(shapes.rb_kernels)
    for i, (u, k) in enumerate(zip(shapes.up_rates, shapes.up_kernels)):
        x = F.leaky_relu(x, 0.1)                                                                  # :118
        x = F.conv_transpose1d(x, fold_weight_norm(sd, f"{prefix}.ups.{i}"), sd[f"{prefix}.ups.{i}.bias"],
                               stride=u, padding=(k - u) // 2)                                    # :119
        if taps is not None:
            taps[f"voc_up{i}"] = x.squeeze(0)
        xs = None
        for j, (rk, dils) in enumerate(zip(shapes.rb_kernels, shapes.rb_dils)):
            r = f"{prefix}.resblocks.{i * nk + j}"
            y = x
            for d_i, d in enumerate(dils):                                                        # :50-57
                xt = F.leaky_relu(y, 0.1)
                xt = F.conv1d(xt, fold_weight_norm(sd, f"{r}.convs1.{d_i}"), sd[f"{r}.convs1.{d_i}.bias"],
                              dilation=d, padding=(rk * d - d) // 2)
                xt = F.leaky_relu(xt, 0.1)
                xt = F.conv1d(xt, fold_weight_norm(sd, f"{r}.convs2.{d_i}"), sd[f"{r}.convs2.{d_i}.bias"],
                              dilation=1, padding=(rk - 1) // 2)
                y = xt + y
            xs = y if xs is None else xs + y                                                      # :121-125
        x = xs / nk                                                                               # :126
        if taps is not None:
            taps[f"voc_mrf{i}"] = x.squeeze(0)
    x = F.leaky_relu(x)                                                                           # :127 (slope 0.01)
    x = F.conv1d(x, fold_weight_norm(sd, prefix + ".conv_post"), sd[prefix + ".conv_post.bias"], padding=3)
    x = torch.tanh(x)                                                                             # :129
    return x.view(-1)


def jets_forward(sd, ling, speaker, style, content, shapes, alpha=1.0, taps: Optional[dict] = None,
                 durations=None, duration_scale=1.0):
    """models/prompt_tts_modified/jets.py:50-71, inference branch (:61-66), one utterance.  ``alpha`` is accepted and, like in
    the reference's inference branch, ignored (see am_forward); ``duration_scale`` is not part of the reference call."""
    with torch.no_grad():
        ling, style, content = _t(ling).long(), _t(style).float(), _t(content).float()
        out = am_forward(sd, ling, int(speaker), style, content, shapes, alpha, taps, durations, duration_scale)
        wav = hifigan_forward(sd, out["dec_outputs"].t().contiguous(), shapes, taps)              # :62-66
        if taps is not None:
            taps["wav"] = wav
        out["wav_predictions"] = wav
    return out


def wav_to_int16(wav_f32: np.ndarray) -> np.ndarray:
    """inference_am_vocoder_joint.py:130-131: (x * 32768.0).astype('int16') -- C cast,
    truncation toward zero; out-of-range wraps like numpy's float->int64->int16 path."""
    scaled = np.asarray(wav_f32, np.float32) * np.float32(32768.0)
    return scaled.astype(np.int64).astype(np.int16)
