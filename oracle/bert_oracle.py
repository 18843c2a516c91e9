"""fp32 CPU restatement of the SimBERT prompt / content encoder's forward: TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference computes its style / content embeddings as ``StyleEncoder.forward(...)["pooled_output"]``
(/root/reference/models/prompt_tts_modified/simbert.py:48-72; callers inference_am_vocoder_joint.py:25-38,
predict.py:142-158), where ``StyleEncoder.bert = AutoModel.from_pretrained(config.bert_path)`` (simbert.py:37) is a
``transformers`` BertModel -- a third-party dependency that is not under /root/reference (pinned transformers==4.26.1 in the
reference's cog.yaml:23 / setup.py:18; 5.15.0 is installed in this image) with weights WangZeJun/simbert-base-chinese that are
a download.  This file restates the published BERT algorithm (Devlin et al. 2018; transformers modeling_bert.py:
BertEmbeddings, BertSelfAttention, BertSelfOutput, BertIntermediate, BertOutput, BertPooler) as plain functions over a state dict.

Pinning: tests/test_oracle_bert.py checks this restatement (a) against ``transformers.BertModel`` itself with the seeded synthetic
weights loaded (strict) and (b) against tests/golden/simbert_*.npz, produced by tests/golden/make_golden_simbert.py by running
the reference's own ``StyleEncoder.forward`` with those weights (AutoModel.from_pretrained patched to build the model from a
config, because the hub is unreachable).  The real SimBERT weights are not available offline: parity with the released model is
pinned on architecture + arithmetic, not on its weights.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def bert_pooled_output(sd, input_ids, token_type_ids=None, heads: int = 12, eps: float = 1e-12, prefix: str = "bert.", taps=None):
    """One text (B = 1, attention_mask all ones -- how every reference call site invokes it).  input_ids (N,) int64.
    Returns pooled_output (hidden,) fp32."""
    with torch.no_grad():
        g = lambda k: _t(sd[prefix + k]).float()  # noqa: E731
        ids = _t(input_ids).long()
        N = ids.shape[0]
        tt = torch.zeros(N, dtype=torch.long) if token_type_ids is None else _t(token_type_ids).long()
        # BertEmbeddings: (word + token_type) + position -> LayerNorm (dropout = identity in eval)
        x = g("embeddings.word_embeddings.weight")[ids] + g("embeddings.token_type_embeddings.weight")[tt]
        x = x + g("embeddings.position_embeddings.weight")[torch.arange(N)]
        H = x.shape[1]
        x = F.layer_norm(x, (H,), g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
        if taps is not None:
            taps["emb"] = x
        dk = H // heads
        i = 0
        while prefix + f"encoder.layer.{i}.attention.self.query.weight" in sd:
            p = f"encoder.layer.{i}."
            q = F.linear(x, g(p + "attention.self.query.weight"), g(p + "attention.self.query.bias")).view(N, heads, dk).transpose(0, 1)
            k = F.linear(x, g(p + "attention.self.key.weight"), g(p + "attention.self.key.bias")).view(N, heads, dk).transpose(0, 1)
            v = F.linear(x, g(p + "attention.self.value.weight"), g(p + "attention.self.value.bias")).view(N, heads, dk).transpose(0, 1)
            att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dk), dim=-1)          # BertSelfAttention
            ctx = torch.matmul(att, v).transpose(0, 1).reshape(N, H)
            y = F.linear(ctx, g(p + "attention.output.dense.weight"), g(p + "attention.output.dense.bias"))
            x = F.layer_norm(y + x, (H,), g(p + "attention.output.LayerNorm.weight"), g(p + "attention.output.LayerNorm.bias"), eps)   # BertSelfOutput
            h = F.gelu(F.linear(x, g(p + "intermediate.dense.weight"), g(p + "intermediate.dense.bias")))                          # BertIntermediate (erf gelu)
            y = F.linear(h, g(p + "output.dense.weight"), g(p + "output.dense.bias"))
            x = F.layer_norm(y + x, (H,), g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)                   # BertOutput
            if taps is not None:
                taps[f"layer{i}"] = x
            i += 1
        return torch.tanh(F.linear(x[0], g("pooler.dense.weight"), g("pooler.dense.bias")))              # BertPooler on [CLS]
