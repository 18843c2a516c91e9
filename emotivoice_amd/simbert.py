"""Drop-in mirror of the reference's ``StyleEncoder`` (models/prompt_tts_modified/simbert.py:33-72) on top of libevhip.so.

The reference runs this BERT-base on the CPU, twice per utterance (prompt and content text:
inference_am_vocoder_joint.py:25-38,106-107); once the generator takes ~3 ms per utterance that CPU forward is >95 % of the
end-to-end latency.  ``StyleEncoderHIP`` keeps the object protocol the callers use --
``StyleEncoder(config)``, ``load_state_dict(ckpt, strict=False)``, ``encoder(input_ids=..., token_type_ids=...,
attention_mask=...)["pooled_output"]`` -- and computes ``pooled_output`` with ev_style_embed.  The four classification heads
and ``style_embed_proj`` only exist for the style-encoder pre-training loss and never reach ``pooled_output``; their outputs
are returned as None.  The tokenizer is a host-side WordPiece lookup as in the reference: ``emotivoice_amd.wordpiece`` from the
checkpoint directory's ``vocab.txt`` (tested against transformers' BERT tokenizer), or any object with the same call protocol.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .engine import EVEngine, EVError
from .packer import pack_bert_state_dict


class StyleEncoderHIP:
    def __init__(self, config=None, device: str = "cuda:0", engine: Optional[EVEngine] = None):
        """``config`` is the reference's Config object (only ``bert_hidden_size`` is read, when present)."""
        self.config = config
        self._device_id = int(str(device).split(":")[1]) if ":" in str(device) else 0
        self._engine = engine
        self._own_engine = engine is None
        self._loaded = False
        hidden = getattr(config, "bert_hidden_size", 768) if config is not None else 768
        if hidden % 128 or hidden // 64 * 64 != hidden:
            raise ValueError("bert_hidden_size must be a multiple of 128 with 64-wide heads")

    def to(self, device):
        s = str(device)
        if s == "cpu":
            raise EVError("StyleEncoderHIP has no CPU path: it needs a HIP device (cuda:N)")
        self._device_id = int(s.split(":")[1]) if ":" in s else 0
        return self

    def eval(self):
        return self

    def _eng(self) -> EVEngine:
        if self._engine is None:
            self._engine = EVEngine(device_id=self._device_id)
        return self._engine

    def load_state_dict(self, state_dict, strict: bool = False):
        """Accepts the StyleEncoder checkpoint's ``model`` dict (``bert.*`` + head keys; a leading ``module.`` is stripped like
        predict.py:113-117 does) or a bare BertModel state dict."""
        blob, _, cfg = pack_bert_state_dict(state_dict)
        self._eng().style_load(blob, cfg)
        self.hidden = cfg["hidden"]
        self._loaded = True
        return self

    def close(self):
        if self._engine is not None and self._own_engine:
            self._engine.close()
        self._engine = None

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None):
        if not self._loaded:
            raise EVError("load_state_dict() first")
        is_torch = hasattr(input_ids, "detach")
        to_np = lambda x: (x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x))  # noqa: E731
        ids = to_np(input_ids).astype(np.int64)
        if ids.ndim == 1:
            ids = ids[None]
        B, N = ids.shape
        mask = np.ones((B, N), np.int64) if attention_mask is None else to_np(attention_mask).astype(np.int64).reshape(B, N)
        lens = mask.sum(1)
        # right-padded batches only (what a HF tokenizer produces): the valid tokens of a text are its first lens[b] ids
        if not all((mask[b, :lens[b]] == 1).all() for b in range(B)) or lens.min() <= 0:
            raise ValueError("attention_mask must be a right-padded prefix mask with at least one token per text")
        tt = None
        if token_type_ids is not None:
            t = to_np(token_type_ids).astype(np.int64).reshape(B, N)
            tt = [t[b, :lens[b]] for b in range(B)]
        pooled = self._eng().style_embed([ids[b, :lens[b]] for b in range(B)], tt)
        if is_torch:
            import torch
            pooled = torch.from_numpy(pooled).to(input_ids.device)
        return {"pooled_output": pooled, "pitch_outputs": None, "speed_outputs": None, "energy_outputs": None, "emotion_outputs": None}
