#!/usr/bin/env python3
"""Generate tests/golden/simbert_*.npz by running the REFERENCE's own StyleEncoder.forward
(/root/reference/models/prompt_tts_modified/simbert.py:48-72), imported in place -- only possible in the build container.

``StyleEncoder.__init__`` calls ``AutoModel.from_pretrained(config.bert_path)`` (simbert.py:37), i.e. a hub download of
WangZeJun/simbert-base-chinese that is unreachable here; ``AutoModel.from_pretrained`` is patched for the duration of the
constructor to build the same architecture (transformers BertModel, BERT-base geometry of that checkpoint) from a config.  The
seeded synthetic weights of emotivoice_amd/synthetic.py are then loaded with ``load_state_dict`` exactly like the reference
loads its style-encoder checkpoint (inference_am_vocoder_joint.py:61-67, strict=False because the classification heads are not
in the synthetic dict), and forward() runs as in get_style_embedding (:25-38): one text per call, attention mask all ones.

Usage: python tests/golden/make_golden_simbert.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from emotivoice_amd.synthetic import synth_bert_state_dict, synth_token_ids  # noqa: E402

GEOM = dict(vocab_size=13685, hidden=768, layers=12, intermediate=3072, max_position=512, type_vocab=2)
CASES = {"simbert_prompt8": (0, 31, [8]), "simbert_content57": (0, 32, [57])}


class _Cfg:        # the attributes StyleEncoder.__init__ reads (config/joint/config.py)
    bert_path = "WangZeJun/simbert-base-chinese"
    bert_hidden_size = 768
    style_dim = 128
    pitch_n_labels = speed_n_labels = energy_n_labels = emotion_n_labels = 3


def build_reference_style_encoder():
    import transformers
    from transformers import BertConfig, BertModel
    import models.prompt_tts_modified.simbert as simbert
    cfg = BertConfig(vocab_size=GEOM["vocab_size"], hidden_size=GEOM["hidden"], num_hidden_layers=GEOM["layers"], num_attention_heads=12,
                     intermediate_size=GEOM["intermediate"], max_position_embeddings=GEOM["max_position"], type_vocab_size=GEOM["type_vocab"],
                     layer_norm_eps=1e-12, hidden_act="gelu")
    orig = simbert.AutoModel.from_pretrained
    simbert.AutoModel.from_pretrained = staticmethod(lambda *_a, **_k: BertModel(cfg))
    try:
        enc = simbert.StyleEncoder(_Cfg())
    finally:
        simbert.AutoModel.from_pretrained = orig
    return enc.eval(), transformers.__version__


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    enc, ver = build_reference_style_encoder()
    sds = {}
    for name, (wseed, iseed, lens) in CASES.items():
        if wseed not in sds:
            sds[wseed] = synth_bert_state_dict(wseed, **GEOM)
            missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in sds[wseed].items()}, strict=False)
            assert not missing.unexpected_keys and all(".bert." not in k and not k.startswith("bert.") for k in missing.missing_keys), missing
        ids = synth_token_ids(iseed, lens, GEOM["vocab_size"])[0]
        with torch.no_grad():
            out = enc(input_ids=torch.from_numpy(ids)[None], token_type_ids=torch.zeros(1, len(ids), dtype=torch.long),
                      attention_mask=torch.ones(1, len(ids), dtype=torch.long))
        pooled = out["pooled_output"].squeeze(0).numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), input_ids=ids, pooled_output=pooled, weight_seed=np.int64(wseed),
                            transformers_version=np.array(ver))
        print(name, "N", len(ids), "|pooled| max %.3f rms %.3f" % (np.abs(pooled).max(), np.sqrt((pooled ** 2).mean())))


if __name__ == "__main__":
    main()
