"""Pin the SimBERT oracle (oracle/bert_oracle.py): against transformers' BertModel itself (the third-party module the reference's
StyleEncoder wraps, simbert.py:37) and against the fixtures produced by running the reference's StyleEncoder.forward
(tests/golden/make_golden_simbert.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, rel_l2
from emotivoice_amd.synthetic import synth_bert_state_dict, synth_token_ids
from oracle.bert_oracle import bert_pooled_output

FIX = sorted(glob.glob(os.path.join(GOLDEN_DIR, "simbert_*.npz")))


def test_fixtures_present():
    assert len(FIX) >= 2


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_oracle_matches_reference_style_encoder(path):
    g = np.load(path)
    sd = synth_bert_state_dict(int(g["weight_seed"]))
    got = bert_pooled_output(sd, g["input_ids"]).numpy()
    assert got.shape == g["pooled_output"].shape == (768,)
    assert rel_l2(got, g["pooled_output"]) < 2e-5


def test_oracle_matches_transformers_bert_model():
    transformers = pytest.importorskip("transformers")
    geom = dict(vocab_size=997, hidden=768, layers=3, intermediate=3072, max_position=64, type_vocab=2)
    sd = synth_bert_state_dict(5, prefix="", **geom)
    cfg = transformers.BertConfig(vocab_size=997, hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=3072,
                                  max_position_embeddings=64, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    m = transformers.BertModel(cfg).eval()
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys and all("position_ids" in k for k in missing.missing_keys), missing
    ids = synth_token_ids(3, [21], 997)[0]
    tt = np.array([0] * 11 + [1] * 10, np.int64)
    with torch.no_grad():
        ref = m(input_ids=torch.from_numpy(ids)[None], token_type_ids=torch.from_numpy(tt)[None],
                attention_mask=torch.ones(1, 21, dtype=torch.long))["pooler_output"][0].numpy()
    got = bert_pooled_output(sd, ids, tt, prefix="").numpy()
    assert rel_l2(got, ref) < 2e-5


def test_packer_accepts_reference_checkpoint_key_styles():
    """StyleEncoder checkpoints carry 'module.bert....' keys (stripped by key[7:] at predict.py:113-117); a bare BertModel has no
    prefix.  All three spellings must pack to the same blob."""
    from emotivoice_amd.packer import pack_bert_state_dict
    geom = dict(vocab_size=211, hidden=768, layers=1, intermediate=3072, max_position=32, type_vocab=2)
    sd = synth_bert_state_dict(7, **geom)
    a, _, cfg = pack_bert_state_dict(sd)
    b, _, _ = pack_bert_state_dict({"module." + k: v for k, v in sd.items()})
    c, _, _ = pack_bert_state_dict({k[5:]: v for k, v in sd.items()})
    extra = dict(sd)
    extra["pitch_clf.classifier.weight"] = np.zeros((3, 768), np.float32)       # heads present in the checkpoint, unused
    d, _, _ = pack_bert_state_dict(extra)
    assert a == b == c == d
    assert cfg == dict(vocab_size=211, hidden=768, layers=1, intermediate=3072, max_position=32, type_vocab=2)
