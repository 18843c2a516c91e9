"""MI355X: the SimBERT prompt / content encoder on the device (ev_style_embed) against the fixtures produced by the reference's
own StyleEncoder.forward and against the CPU oracle; ragged batches must reproduce the per-text (B = 1) results."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, rel_l2

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FIX = sorted(glob.glob(os.path.join(GOLDEN_DIR, "simbert_*.npz")))
TOL = 2e-5          # fp32-class arithmetic end to end (measured ~1e-6): the pooled output conditions the bit-exact duration path


@pytest.fixture(scope="module")
def enc():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from emotivoice_amd.simbert import StyleEncoderHIP
    from emotivoice_amd.synthetic import synth_bert_state_dict
    e = StyleEncoderHIP(None).to("cuda:0")
    e.load_state_dict(synth_bert_state_dict(0))
    return e.eval()


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_reference_style_encoder_fixture(enc, path):
    g = np.load(path)
    ids = torch.from_numpy(g["input_ids"])[None]
    out = enc(input_ids=ids, token_type_ids=torch.zeros_like(ids), attention_mask=torch.ones_like(ids))
    pooled = out["pooled_output"].cpu().squeeze().numpy()          # the callers' own post-processing (:37)
    assert pooled.shape == (768,)
    assert rel_l2(pooled, g["pooled_output"]) < TOL


def test_ragged_batch_equals_per_text_oracle(enc):
    from emotivoice_amd.synthetic import synth_bert_state_dict, synth_token_ids
    from oracle.bert_oracle import bert_pooled_output
    sd = synth_bert_state_dict(0)
    lens = [5, 64, 1, 130, 17, 257]
    ids = synth_token_ids(77, lens)
    rng = np.random.default_rng(1)
    tts = [rng.integers(0, 2, size=n, dtype=np.int64) for n in lens]
    N = max(lens)
    pad_ids = np.zeros((len(lens), N), np.int64)
    pad_tt = np.zeros((len(lens), N), np.int64)
    mask = np.zeros((len(lens), N), np.int64)
    for b, n in enumerate(lens):
        pad_ids[b, :n], pad_tt[b, :n], mask[b, :n] = ids[b], tts[b], 1
    got = enc(input_ids=pad_ids, token_type_ids=pad_tt, attention_mask=mask)["pooled_output"]
    assert got.shape == (len(lens), 768)
    for b, n in enumerate(lens):
        ref = bert_pooled_output(sd, ids[b], tts[b]).numpy()
        assert rel_l2(got[b], ref) < TOL, (b, n)
        solo = enc(input_ids=ids[b][None], token_type_ids=tts[b][None])["pooled_output"][0]
        assert np.array_equal(solo, got[b]), b                       # batch-invariant, bit for bit


def test_longest_text_512_tokens(enc):
    """max_position_embeddings = 512 tokens (the longest text the reference's tokenizer + BERT accept) against the oracle."""
    from emotivoice_amd.synthetic import synth_bert_state_dict, synth_token_ids
    from oracle.bert_oracle import bert_pooled_output
    ids = synth_token_ids(9, [512])[0]
    got = enc(input_ids=ids[None])["pooled_output"][0]
    assert rel_l2(got, bert_pooled_output(synth_bert_state_dict(0), ids).numpy()) < TOL


def test_errors(enc):
    from emotivoice_amd.engine import EVError
    with pytest.raises(EVError):
        enc(input_ids=np.array([[101, 999999, 102]]))               # id outside the vocabulary
    with pytest.raises(EVError):
        enc(input_ids=np.ones((1, 600), np.int64))                  # longer than max_position_embeddings


class _StubTokenizer:
    """What the callers need from AutoTokenizer.from_pretrained(config.bert_path) (predict.py:145-149): text list -> dict of
    (1, N) id / type / mask arrays with [CLS] ... [SEP].  (The WordPiece vocabulary is a download; any deterministic map serves.)"""

    def __call__(self, texts, return_tensors="np"):
        import zlib
        ids = [101] + [103 + zlib.crc32(w.encode()) % 13000 for w in texts[0].split()] + [102]
        a = np.array([ids], np.int64)
        return {"input_ids": a, "token_type_ids": np.zeros_like(a), "attention_mask": np.ones_like(a)}


def test_predictor_with_style_encoder_on_device(tmp_path):
    """predict.py flow end to end on the GPU: tokenizer -> StyleEncoder (device) -> pooled_output -> generator -> wav; the embeddings
    the generator received are the oracle's BERT outputs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import wave
    from emotivoice_amd.predict import Predictor
    from emotivoice_amd.synthetic import synth_bert_state_dict, synth_state_dict
    from oracle.bert_oracle import bert_pooled_output
    toks = ["_", "<sos/eos>"] + ["p%d" % i for i in range(500)]
    (tmp_path / "tokenlist").write_text("\n".join(toks) + "\n")
    (tmp_path / "speaker2").write_text("\n".join(["8051"] + ["s%d" % i for i in range(2013)]))
    bsd = synth_bert_state_dict(0)
    ckpt_style = {"module." + k: v for k, v in bsd.items()}            # the style-encoder checkpoint's key style (predict.py:113-117)
    p = Predictor(str(tmp_path / "tokenlist"), str(tmp_path / "speaker2"), str(tmp_path / "out"),
                  g2p={"English": lambda text: "<sos/eos> " + " ".join("p%d" % (len(w) % 400) for w in text.split()) + " <sos/eos>"})
    p.setup_models(generator_state_dict=synth_state_dict(0, "parity"), style_encoder_state_dict=ckpt_style, tokenizer=_StubTokenizer())
    emb = p.get_style_embedding("Happy and loud")
    ids = _StubTokenizer()(["Happy and loud"])["input_ids"][0]
    assert rel_l2(emb, bert_pooled_output(bsd, ids).numpy()) < TOL
    path = p.predict(prompt="Happy", content="a small test sentence for the predictor", language="English", speaker="8051")
    with wave.open(path) as w:
        assert w.getframerate() == 16000 and w.getnframes() % 256 == 0 and w.getnframes() > 0
    with pytest.raises(ValueError):
        p.predict(content="中文 text", language="English", speaker="8051")
    with pytest.raises(ValueError):
        p.predict(content="english only", language="Chinese", speaker="8051")
