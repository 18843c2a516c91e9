"""Weight packer: reference ``JETSGenerator`` state dict -> one flat blob for libevhip.so.

Replaces what ``load_state_dict`` + the per-forward weight-norm recompute do in the reference
(inference_am_vocoder_joint.py:72-73; models/hifigan/models.py:10-14 -- the joint path never calls
``remove_weight_norm`` so the reference re-evaluates ``g*v/||v||`` for 50 convs on every forward,
SURVEY.md section 6).  Here weight-norm is folded once, weights are laid out as the GEMM kernels read
them ([N][taps][K], K contiguous), q/k/v are fused, ``embed_projection1`` is split into its
time-varying 384 columns and the per-utterance conditioning columns, ConvTranspose1d becomes a
3-tap polyphase conv, and the sinusoid table (not in the state dict, modules/encoder.py:213-237) is
regenerated.  The blob is what one RCCL broadcast ships to the other ranks.

Blob layout (little endian):
  "EVW1\\0\\0\\0\\0" | u32 count | u32 0 | count x entry | data
  entry = char name[64] | u32 dtype (0 f16, 1 f32) | u32 ndim | u64 dims[4] | u64 offset | u64 nbytes
Tensor data is 256-byte aligned.
"""
from __future__ import annotations

import json
import struct
from typing import Dict, Tuple

import numpy as np

MEL_PAD = 96
DT_F16, DT_F32 = 0, 1
_ENTRY = struct.Struct("<64sII4QQQ")


def _np(x) -> np.ndarray:
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _fold_weight_norm(sd, prefix) -> np.ndarray:
    """w = g * v / ||v||, norm over all dims but 0 (torch weight_norm default dim=0).  Accepts the
    torch>=2.1 parametrization keys, the legacy weight_g/weight_v keys of the released checkpoints
    (trained with torch 1.11 / 2.0.1, reference Dockerfile:6, cog.yaml:14) and plain weights."""
    for gk, vk in ((".parametrizations.weight.original0", ".parametrizations.weight.original1"),
                   (".weight_g", ".weight_v")):
        if prefix + gk in sd:
            g = _np(sd[prefix + gk]).astype(np.float64)
            v = _np(sd[prefix + vk]).astype(np.float64)
            nrm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(1)).reshape((-1,) + (1,) * (v.ndim - 1))
            return (g * v / nrm).astype(np.float32)
    return _np(sd[prefix + ".weight"]).astype(np.float32)


def _conv_to_gemm(w: np.ndarray, k_pad: int | None = None) -> np.ndarray:
    """Conv1d weight [C_out, C_in, k] -> [C_out][k][C_in] (optionally zero-padding C_in)."""
    w = np.ascontiguousarray(np.transpose(w, (0, 2, 1)))
    if k_pad is not None and k_pad != w.shape[2]:
        out = np.zeros((w.shape[0], w.shape[1], k_pad), w.dtype)
        out[:, :, : w.shape[2]] = w
        w = out
    return w


def _convT_to_gemm(wt: np.ndarray, stride: int) -> np.ndarray:
    """ConvTranspose1d weight [C_in, C_out, K=2s], padding s/2 -> 3-tap conv [s*C_out][3][C_in].

    out[t*s + p, co] = sum_ci sum_tau x[t + tau - 1, ci] * Wt[ci, co, kk(p, tau)] with q = p + s/2:
      tau=0 (x[t-1]): kk = q + s   (only if q <  s)
      tau=1 (x[t]  ): kk = q
      tau=2 (x[t+1]): kk = q - s   (only if q >= s)
    (from n = i*s - pad + kk of torch.nn.ConvTranspose1d; models/hifigan/models.py:99-103)."""
    cin, cout, K = wt.shape
    s = stride
    assert K == 2 * s and s % 2 == 0
    out = np.zeros((s * cout, 3, cin), np.float32)
    for p in range(s):
        q = p + s // 2
        blk = out[p * cout:(p + 1) * cout]
        if q < s:
            blk[:, 0, :] = wt[:, :, q + s].T
        blk[:, 1, :] = wt[:, :, q].T
        if q >= s:
            blk[:, 2, :] = wt[:, :, q - s].T
    return out


def sinusoid_table(length: int, d: int) -> np.ndarray:
    """modules/encoder.py:216-237, evaluated with torch fp32 ops like the reference does."""
    import math

    import torch

    pos = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(length, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.numpy()


def _pe_div_term(d: int) -> np.ndarray:
    """div_term of modules/encoder.py:224-227 with the reference's fp32 op order (table extension on the device)."""
    import math

    import torch

    return torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d)).numpy()


class _Blob:
    def __init__(self):
        self.items = []

    def add(self, name: str, arr: np.ndarray, dtype: int):
        arr = np.ascontiguousarray(arr, dtype=np.float16 if dtype == DT_F16 else np.float32)
        assert arr.ndim <= 4 and len(name) < 64, name
        self.items.append((name, arr, dtype))

    def both(self, name: str, arr: np.ndarray):
        self.add(name + "32", arr, DT_F32)
        self.add(name + "16", arr, DT_F16)

    def split(self, name: str, arr: np.ndarray):
        """fp32 weight as an fp16 hi/lo pair for the split-precision GEMM (ev_gemm.hip: conv_gemm_split_kernel):
        w = hi + 2^-11 * lo with hi = fp16(w), lo = fp16((w - hi) * 2^11)."""
        w = np.ascontiguousarray(arr, np.float32)
        hi = w.astype(np.float16)
        lo = ((w - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        self.add(name + "32h", hi, DT_F16)
        self.add(name + "32l", lo, DT_F16)

    def lo(self, name: str, arr: np.ndarray):
        """low part only (the high part fp16(w) is the tensor's plain fp16 copy): fp16((w - fp16(w)) * 2^11)."""
        w = np.ascontiguousarray(arr, np.float32)
        hi = w.astype(np.float16)
        self.add(name, ((w - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16), DT_F16)

    def raw(self, name: str, data: np.ndarray):
        """opaque bytes (the device code defines the layout), stored as an fp16-typed entry of ceil(n / 2) elements."""
        d = np.ascontiguousarray(data, np.uint8).reshape(-1)
        if d.size % 2:
            d = np.concatenate([d, np.zeros(1, np.uint8)])
        self.items.append((name, d.view(np.float16), DT_F16))

    def mx(self, name: str, w: np.ndarray):
        """fp4 planes of a GEMM-layout weight [N][taps][K] for the "mx" precision (mxfp4.pack_weight_planes), only for the shapes
        conv_gemm_mx_kernel takes: N and K multiples of 128 and 3 / 7 / 11 taps.  The engine selects the MX kernel per layer by the
        presence of this entry."""
        N, taps, K = w.shape
        if N % 128 == 0 and K % 128 == 0 and taps in (3, 7, 11):
            from .mxfp4 import pack_weight_planes
            self.raw(name, pack_weight_planes(w))

    def mx1(self, name: str, w: np.ndarray):
        """fp4 planes of an nn.Linear weight [N][1][K] (N, K multiples of 128) for the one-tap MX GEMM (gemm_mx1_kernel): the same plane layout."""
        N, taps, K = w.shape
        if N % 128 == 0 and K % 128 == 0 and taps == 1:
            from .mxfp4 import pack_weight_planes
            self.raw(name, pack_weight_planes(w))

    def mx_pair(self, name: str, w: np.ndarray):
        """fp4 planes of a C = 32 ResBlock conv for the fused MX pair kernel (mxfp4.pack_pair_weight_planes)."""
        N, taps, K = w.shape
        if N == 32 and K == 32 and taps in (3, 7, 11):
            from .mxfp4 import pack_pair_weight_planes
            self.raw(name, pack_pair_weight_planes(w))

    def mx_c64(self, name: str, w: np.ndarray):
        """fp4 planes of a C = 64 conv (ResBlock convs of stage 2, the last up-conv) for conv_c64_mx_kernel (mxfp4.pack_c64_weight_planes)."""
        N, taps, K = w.shape
        if N == 64 and K == 64 and taps in (3, 7, 11):
            from .mxfp4 import pack_c64_weight_planes
            self.raw(name, pack_c64_weight_planes(w))

    def finish(self) -> Tuple[bytes, dict]:
        n = len(self.items)
        off = 16 + n * _ENTRY.size
        table, manifest = [], {}
        chunks = []
        for name, arr, dtype in self.items:
            off = (off + 255) // 256 * 256
            dims = list(arr.shape) + [0] * (4 - arr.ndim)
            if arr.ndim == 0:
                dims = [1, 0, 0, 0]
            table.append(_ENTRY.pack(name.encode(), dtype, max(arr.ndim, 1), *dims, off, arr.nbytes))
            manifest[name] = dict(dtype="f16" if dtype == DT_F16 else "f32", shape=list(arr.shape), offset=off,
                                  nbytes=arr.nbytes)
            chunks.append((off, arr))
            off += arr.nbytes
        buf = bytearray(off)
        buf[0:8] = b"EVW1\0\0\0\0"
        struct.pack_into("<II", buf, 8, n, 0)
        pos = 16
        for t in table:
            buf[pos:pos + len(t)] = t
            pos += len(t)
        for o, arr in chunks:
            buf[o:o + arr.nbytes] = arr.tobytes()
        return bytes(buf), manifest


def pack_state_dict(sd: Dict[str, object], shapes=None, pe_len: int = 4096) -> Tuple[bytes, str]:
    """Pack a reference ``JETSGenerator`` state dict (``am.*`` / ``generator.*`` keys, optionally with a
    DDP ``module.`` prefix) into (blob, manifest_json)."""
    from .config import EVShapes

    s = shapes or EVShapes()
    if any(k.startswith("module.") for k in sd):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    H = s.hidden
    b = _Blob()
    f32 = lambda k: _np(sd[k]).astype(np.float32)  # noqa: E731

    b.add("pe", sinusoid_table(pe_len, H), DT_F32)
    b.add("pe_div", _pe_div_term(H), DT_F32)       # lets the engine extend the table on the device for longer utterances
    b.add("tok_emb", f32("am.src_word_emb.weight"), DT_F32)
    b.add("spk_emb", f32("am.spk_tokenizer.weight"), DT_F32)
    for short, pre, nl in (("enc", "am.encoder", s.enc_layers), ("dec", "am.decoder", s.dec_layers)):
        b.add(f"{short}.alpha", f32(f"{pre}.embed.0.alpha").reshape(1), DT_F32)
        for i in range(nl):
            p = f"{pre}.encoders.{i}"
            wq = [f32(f"{p}.self_attn.linear_{n}.weight") for n in "qkv"]
            bq = [f32(f"{p}.self_attn.linear_{n}.bias") for n in "qkv"]
            b.both(f"{short}.{i}.qkv.w", np.concatenate(wq, 0)[:, None, :])
            if short == "enc":       # token-rate stack: also as hi/lo split for the split-precision fp32 GEMM
                b.split(f"{short}.{i}.qkv.w", np.concatenate(wq, 0)[:, None, :])
                b.split(f"{short}.{i}.out.w", f32(f"{p}.self_attn.linear_out.weight")[:, None, :])
                b.split(f"{short}.{i}.ffn1.w", _conv_to_gemm(f32(f"{p}.feed_forward.w_1.weight")))
                b.split(f"{short}.{i}.ffn2.w", _conv_to_gemm(f32(f"{p}.feed_forward.w_2.weight")))
            else:                    # mel decoder, split-precision mode (EV_PREC_X3): hi = the w16 copy, lo packed here
                b.lo(f"{short}.{i}.qkv.w32l", np.concatenate(wq, 0)[:, None, :])
                b.lo(f"{short}.{i}.out.w32l", f32(f"{p}.self_attn.linear_out.weight")[:, None, :])
                b.lo(f"{short}.{i}.ffn1.w32l", _conv_to_gemm(f32(f"{p}.feed_forward.w_1.weight")))
                b.mx(f"{short}.{i}.ffn1.wmx", _conv_to_gemm(f32(f"{p}.feed_forward.w_1.weight")))       # "mx" decoder: conv-FFN on the MX kernel
                b.mx(f"{short}.{i}.ffn2.wmx", _conv_to_gemm(f32(f"{p}.feed_forward.w_2.weight")))
                b.mx1(f"{short}.{i}.qkv.wmx", np.concatenate(wq, 0)[:, None, :])                           # ... and the QKV / output projections on the one-tap MX GEMM
                b.mx1(f"{short}.{i}.out.wmx", f32(f"{p}.self_attn.linear_out.weight")[:, None, :])
                b.lo(f"{short}.{i}.ffn2.w32l", _conv_to_gemm(f32(f"{p}.feed_forward.w_2.weight")))
            b.add(f"{short}.{i}.qkv.b", np.concatenate(bq, 0), DT_F32)
            b.both(f"{short}.{i}.out.w", f32(f"{p}.self_attn.linear_out.weight")[:, None, :])
            b.add(f"{short}.{i}.out.b", f32(f"{p}.self_attn.linear_out.bias"), DT_F32)
            b.both(f"{short}.{i}.ffn1.w", _conv_to_gemm(f32(f"{p}.feed_forward.w_1.weight")))
            b.add(f"{short}.{i}.ffn1.b", f32(f"{p}.feed_forward.w_1.bias"), DT_F32)
            b.both(f"{short}.{i}.ffn2.w", _conv_to_gemm(f32(f"{p}.feed_forward.w_2.weight")))
            b.add(f"{short}.{i}.ffn2.b", f32(f"{p}.feed_forward.w_2.bias"), DT_F32)
            for j in (1, 2):
                b.add(f"{short}.{i}.ln{j}.g", f32(f"{p}.norm{j}.weight"), DT_F32)
                b.add(f"{short}.{i}.ln{j}.b", f32(f"{p}.norm{j}.bias"), DT_F32)
        b.add(f"{short}.after.g", f32(f"{pre}.after_norm.weight"), DT_F32)
        b.add(f"{short}.after.b", f32(f"{pre}.after_norm.bias"), DT_F32)

    wp = f32("am.embed_projection1.weight")              # [H, H + H + 2*bert], column order x|spk|style|content
    b.add("proj.w32", wp[:, None, :H], DT_F32)
    b.split("proj.w", wp[:, None, :H])
    b.add("proj.wcond", wp[:, H:], DT_F32)
    b.add("proj.b", f32("am.embed_projection1.bias"), DT_F32)
    for short, pre, nl in (("dur", "am.duration_predictor", s.dur_layers), ("pitch", "am.pitch_predictor", s.pitch_layers),
                           ("energy", "am.energy_predictor", s.energy_layers)):
        for i in range(nl):
            b.add(f"{short}.{i}.conv.w32", _conv_to_gemm(f32(f"{pre}.conv.{i}.0.weight")), DT_F32)
            b.split(f"{short}.{i}.conv.w", _conv_to_gemm(f32(f"{pre}.conv.{i}.0.weight")))
            b.add(f"{short}.{i}.conv.b", f32(f"{pre}.conv.{i}.0.bias"), DT_F32)
            b.add(f"{short}.{i}.ln.g", f32(f"{pre}.conv.{i}.2.weight"), DT_F32)
            b.add(f"{short}.{i}.ln.b", f32(f"{pre}.conv.{i}.2.bias"), DT_F32)
        b.add(f"{short}.lin.w", f32(f"{pre}.linear.weight").reshape(H), DT_F32)
        b.add(f"{short}.lin.b", f32(f"{pre}.linear.bias").reshape(1), DT_F32)
    for short, pre in (("pitch_emb", "am.pitch_embed.0"), ("energy_emb", "am.energy_embed.0")):
        b.add(f"{short}.w", np.ascontiguousarray(f32(f"{pre}.weight")[:, 0, :].T), DT_F32)   # [k][C]
        b.add(f"{short}.b", f32(f"{pre}.bias"), DT_F32)
    wm = np.zeros((MEL_PAD, 1, H), np.float32)
    wm[: s.n_mels, 0] = f32("am.to_mel.weight")
    bm = np.zeros(MEL_PAD, np.float32)
    bm[: s.n_mels] = f32("am.to_mel.bias")
    b.both("to_mel.w", wm)
    b.lo("to_mel.w32l", wm)
    b.add("to_mel.b", bm, DT_F32)

    g = "generator"
    # every generator conv: fp16 weights ("w16", also the hi part of the split) + the lo part ("w16l") for the split-precision mode
    wpre = _conv_to_gemm(_fold_weight_norm(sd, f"{g}.conv_pre"), MEL_PAD)
    b.add("voc.pre.w16", wpre, DT_F16)
    b.lo("voc.pre.w16l", wpre)
    b.add("voc.pre.b", f32(f"{g}.conv_pre.bias"), DT_F32)
    ch = s.up_init_ch
    nk = len(s.rb_kernels)
    for i, (u, k) in enumerate(zip(s.up_rates, s.up_kernels)):
        assert k == 2 * u, "polyphase packing needs kernel = 2*stride"
        wup = _convT_to_gemm(_fold_weight_norm(sd, f"{g}.ups.{i}"), u)
        b.add(f"voc.up{i}.w16", wup, DT_F16)
        b.lo(f"voc.up{i}.w16l", wup)
        b.mx(f"voc.up{i}.wmx", wup)
        b.mx_c64(f"voc.up{i}.wcmx", wup)
        b.add(f"voc.up{i}.b", np.tile(f32(f"{g}.ups.{i}.bias"), u), DT_F32)
        ch //= 2
        for j in range(nk):
            r = i * nk + j
            for d in range(len(s.rb_dils[j])):
                for grp, short in (("convs1", "c1"), ("convs2", "c2")):
                    pre = f"{g}.resblocks.{r}.{grp}.{d}"
                    wrb = _conv_to_gemm(_fold_weight_norm(sd, pre))
                    b.add(f"voc.rb{r}.{short}.{d}.w16", wrb, DT_F16)
                    b.lo(f"voc.rb{r}.{short}.{d}.w16l", wrb)
                    b.mx(f"voc.rb{r}.{short}.{d}.wmx", wrb)
                    b.mx_pair(f"voc.rb{r}.{short}.{d}.wpmx", wrb)
                    b.mx_c64(f"voc.rb{r}.{short}.{d}.wcmx", wrb)
                    b.add(f"voc.rb{r}.{short}.{d}.b", f32(pre + ".bias"), DT_F32)
    wpost = _fold_weight_norm(sd, f"{g}.conv_post")     # [1, C, 7]
    b.add("voc.post.w", np.ascontiguousarray(wpost[0].T), DT_F32)   # [7][C]
    b.add("voc.post.b", f32(f"{g}.conv_post.bias").reshape(1), DT_F32)
    blob, manifest = b.finish()
    return blob, json.dumps(manifest)


# ----------------------------------------------------------------------------------------------------------------------
# SimBERT prompt / content encoder (reference models/prompt_tts_modified/simbert.py:33-46: StyleEncoder.bert =
# AutoModel.from_pretrained(config.bert_path), a transformers BertModel; the four classification heads and
# style_embed_proj are training-only and do not reach "pooled_output").
def pack_bert_state_dict(sd: Dict[str, object]) -> Tuple[bytes, str, dict]:
    """Pack a StyleEncoder / BertModel state dict for ev_style_load_weights.  Accepts the keys of the reference's StyleEncoder
    (``bert.embeddings...``, optionally behind the DDP ``module.`` prefix its checkpoints carry, predict.py:113-117), or of a bare
    transformers BertModel (``embeddings...``).  Returns (blob, manifest_json, config dict for ev_bert_config).
    GEMM weights are stored as the fp16 hi / lo pair of the split-precision kernel; every name carries the "sb." prefix."""
    if any(k.startswith("module.") for k in sd):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    if any(k.startswith("bert.") for k in sd):
        sd = {k[5:]: v for k, v in sd.items() if k.startswith("bert.")}
    f32 = lambda k: _np(sd[k]).astype(np.float32)  # noqa: E731
    b = _Blob()
    word = f32("embeddings.word_embeddings.weight")
    pos = f32("embeddings.position_embeddings.weight")
    typ = f32("embeddings.token_type_embeddings.weight")
    H = word.shape[1]
    b.add("sb.emb.word", word, DT_F32)
    b.add("sb.emb.pos", pos, DT_F32)
    b.add("sb.emb.type", typ, DT_F32)
    b.add("sb.emb.ln.g", f32("embeddings.LayerNorm.weight"), DT_F32)
    b.add("sb.emb.ln.b", f32("embeddings.LayerNorm.bias"), DT_F32)
    n_layers = 0
    while f"encoder.layer.{n_layers}.attention.self.query.weight" in sd:
        n_layers += 1
    inter = 0

    def gemm(name, w, bias):
        w = np.ascontiguousarray(w, np.float32)[:, None, :]       # [N][taps = 1][K]
        b.add(name + ".w16", w, DT_F16)
        b.lo(name + ".w32l", w)
        b.add(name + ".b", bias, DT_F32)

    for i in range(n_layers):
        p = f"encoder.layer.{i}"
        wq = [f32(f"{p}.attention.self.{n}.weight") for n in ("query", "key", "value")]
        bq = [f32(f"{p}.attention.self.{n}.bias") for n in ("query", "key", "value")]
        gemm(f"sb.{i}.qkv", np.concatenate(wq, 0), np.concatenate(bq, 0))
        gemm(f"sb.{i}.out", f32(f"{p}.attention.output.dense.weight"), f32(f"{p}.attention.output.dense.bias"))
        b.add(f"sb.{i}.ln1.g", f32(f"{p}.attention.output.LayerNorm.weight"), DT_F32)
        b.add(f"sb.{i}.ln1.b", f32(f"{p}.attention.output.LayerNorm.bias"), DT_F32)
        w1 = f32(f"{p}.intermediate.dense.weight")
        inter = w1.shape[0]
        gemm(f"sb.{i}.ffn1", w1, f32(f"{p}.intermediate.dense.bias"))
        gemm(f"sb.{i}.ffn2", f32(f"{p}.output.dense.weight"), f32(f"{p}.output.dense.bias"))
        b.add(f"sb.{i}.ln2.g", f32(f"{p}.output.LayerNorm.weight"), DT_F32)
        b.add(f"sb.{i}.ln2.b", f32(f"{p}.output.LayerNorm.bias"), DT_F32)
    b.add("sb.pool.w", f32("pooler.dense.weight"), DT_F32)
    b.add("sb.pool.b", f32("pooler.dense.bias"), DT_F32)
    blob, manifest = b.finish()
    cfg = dict(vocab_size=int(word.shape[0]), hidden=int(H), layers=n_layers, intermediate=int(inter), max_position=int(pos.shape[0]),
               type_vocab=int(typ.shape[0]))
    return blob, json.dumps(manifest), cfg
