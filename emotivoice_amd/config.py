"""Model hyper-parameters of the hot path.

The reference builds its config in two tiers: a yacs YAML
(config/joint/config.yaml:36-94) patched with ``n_vocab`` / ``n_speaker`` from a Python ``Config``
class (inference_am_vocoder_joint.py:53-58).  ``EVShapes`` holds the subset the inference path
reads; ``from_reference_config`` accepts that same object (any attribute- or dict-style tree), so a
caller that already has the reference's ``conf`` can hand it over unchanged.  yacs is not needed:
``load_yaml`` reads a reference-format config.yaml with pyyaml.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple


@dataclass
class EVShapes:
    n_vocab: int = 502          # config/joint/config.py:56 (len(tokenlist))
    n_speaker: int = 2014       # config/joint/config.py:60 (len(speaker2))
    n_mels: int = 80
    hidden: int = 384
    heads: int = 8
    enc_layers: int = 4
    dec_layers: int = 4
    ffn_kernel: int = 3
    bert_dim: int = 768
    dur_layers: int = 2
    pitch_layers: int = 3
    energy_layers: int = 2      # hard-coded in model_open_source.py:70-76
    var_kernel: int = 3
    var_embed_kernel: int = 9
    up_rates: Tuple[int, ...] = (8, 8, 2, 2)
    up_kernels: Tuple[int, ...] = (16, 16, 4, 4)
    up_init_ch: int = 512
    rb_kernels: Tuple[int, ...] = (3, 7, 11)
    rb_dils: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    sr: int = 16000
    hop: int = 256
    segment_size: int = 32

    @property
    def upsample_factor(self) -> int:
        f = 1
        for u in self.up_rates:
            f *= u
        return f


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _attr(x):
    return AttrDict({k: _attr(v) for k, v in x.items()}) if isinstance(x, dict) else x


def load_yaml(path: str, n_vocab: int = 502, n_speaker: int = 2014) -> AttrDict:
    """Read a reference-format config.yaml (config/joint/config.yaml) without yacs."""
    import yaml

    with open(path) as f:
        conf = _attr(yaml.safe_load(f))
    conf.n_vocab, conf.n_speaker = n_vocab, n_speaker
    return conf


def _get(node, name, default=None):
    if node is None:
        return default
    if isinstance(node, dict):
        return node.get(name, default)
    return getattr(node, name, default)


def from_reference_config(conf) -> EVShapes:
    """Map the reference's ``conf`` (yacs CfgNode / AttrDict / dict) to EVShapes."""
    if conf is None:
        return EVShapes()
    if isinstance(conf, EVShapes):
        return conf
    m = _get(conf, "model")
    d = EVShapes()
    hidden = _get(m, "encoder_n_hidden", d.hidden)
    if _get(m, "decoder_n_hidden", hidden) != hidden or _get(m, "variance_n_hidden", hidden) != hidden:
        raise ValueError("encoder/decoder/variance hidden sizes must match")
    if str(_get(m, "resblock", "1")) != "1":
        raise ValueError("only ResBlock1 generators are supported (config resblock: '1')")
    hop = _get(conf, "hop_length", d.hop)
    s = EVShapes(
        n_vocab=_get(conf, "n_vocab", d.n_vocab), n_speaker=_get(conf, "n_speaker", d.n_speaker),
        n_mels=_get(conf, "n_mels", d.n_mels), hidden=hidden, heads=_get(m, "encoder_n_heads", d.heads),
        enc_layers=_get(m, "encoder_n_layers", d.enc_layers), dec_layers=_get(m, "decoder_n_layers", d.dec_layers),
        ffn_kernel=_get(m, "encoder_kernel_size_conv_mod", d.ffn_kernel), bert_dim=_get(m, "bert_embedding", d.bert_dim),
        dur_layers=_get(m, "duration_n_layers", d.dur_layers), pitch_layers=_get(m, "variance_n_layers", d.pitch_layers),
        energy_layers=2, var_kernel=_get(m, "variance_kernel_size", d.var_kernel),
        var_embed_kernel=_get(m, "variance_embed_kernel_size", d.var_embed_kernel),
        up_rates=tuple(_get(m, "upsample_rates", d.up_rates)), up_kernels=tuple(_get(m, "upsample_kernel_sizes", d.up_kernels)),
        up_init_ch=_get(m, "upsample_initial_channel", d.up_init_ch), rb_kernels=tuple(_get(m, "resblock_kernel_sizes", d.rb_kernels)),
        rb_dils=tuple(tuple(x) for x in _get(m, "resblock_dilation_sizes", d.rb_dils)), sr=_get(conf, "sr", d.sr), hop=hop,
        segment_size=_get(conf, "segment_size", d.segment_size),
    )
    if s.upsample_factor != s.hop:
        raise ValueError("prod(upsample_rates) must equal hop_length")
    return s
