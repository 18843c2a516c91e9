"""emotivoice_amd.wordpiece against transformers' BERT tokenizer (what the reference's AutoTokenizer.from_pretrained(config.bert_path)
resolves to for a BERT checkpoint: inference_am_vocoder_joint.py:25-29,83) on a vocabulary written by the test."""
import numpy as np
import pytest

from emotivoice_amd.wordpiece import WordPieceTokenizer, load_tokenizer

transformers = pytest.importorskip("transformers")

VOCAB = ["[PAD]", "[unused1]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "!", ",", ".", "?", "-", "'", "(", ")", "$", "^", "`", "~", "。", "，", "！", "？", "、",
         "happy", "excited", "sad", "angry", "hap", "##py", "##pi", "##ness", "un", "##happy", "the", "a", "an", "emo", "##ti", "##voice", "voice",
         "multi", "prompt", "controlled", "t", "s", "engine", "e", "##s", "##ing", "##ed", "cafe", "naive", "resume", "uber", "123", "12", "##3", "##45",
         "你", "好", "世", "界", "今", "天", "气", "很", "开", "心", "悲", "伤", "生", "气", "的", "语", "调", "说", "话", "请", "用", "##a", "##b", "b", "c",
         "hello", "world", "##ld", "wor"]
VOCAB = list(dict.fromkeys(VOCAB))          # unique, order kept

TEXTS = [
    "Happy", "Excited", "Sad", "Angry", "happiness", "unhappy thoughts, the end.", "Emoti-Voice - a Multi-Voice and Prompt-Controlled T-T-S Engine",
    "你好，世界！今天天气很好。", "请用开心的语调说话 please", "mixed 你好world hello世界", "café naïve résumé ÜBER", "  leading\tand\ntrailing   spaces  ",
    "price: $12345 (approx.) ~ `code` ^caret", "zzzzunknownzzzz word", "a" * 120 + " a", "hello​world \x00 control\x07chars", "tabs\tand　ideographic space",
    "[MASK] hello [SEP] world", "won't can't it's", "", "！？、。",
]


@pytest.fixture(scope="module")
def toks(tmp_path_factory):
    d = tmp_path_factory.mktemp("bert")
    (d / "vocab.txt").write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    mine = WordPieceTokenizer(str(d / "vocab.txt"))
    ref = transformers.BertTokenizer(str(d / "vocab.txt"), do_lower_case=True)
    return mine, ref, d


@pytest.mark.parametrize("text", TEXTS, ids=[("t%02d" % i) for i in range(len(TEXTS))])
def test_ids_match_transformers(toks, text):
    mine, ref, _ = toks
    want = ref([text])["input_ids"][0]
    assert mine.encode(text) == list(want), (mine.tokenize(text), ref.tokenize(text))


def test_batch_call_is_right_padded_like_the_reference_uses_it(toks):
    mine, ref, d = toks
    out = mine(["你好", "unhappy thoughts, the end."])
    r = ref(["你好", "unhappy thoughts, the end."], padding=True, return_tensors="np")
    for k in ("input_ids", "token_type_ids", "attention_mask"):
        assert out[k].dtype == np.int64 and np.array_equal(out[k], r[k]), k
    one = mine(["Happy"], return_tensors="pt")            # the reference's call: tokenizer([prompt], return_tensors="pt")
    assert tuple(one["input_ids"].shape) == (1, 3) and int(one["attention_mask"].sum()) == 3
    assert isinstance(load_tokenizer(str(d)), WordPieceTokenizer)


def test_cased_vocabulary_option(tmp_path):
    (tmp_path / "vocab.txt").write_text("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "Hello", "hello", "É", "e"]) + "\n", encoding="utf-8")
    (tmp_path / "tokenizer_config.json").write_text('{"do_lower_case": false}')
    mine = load_tokenizer(str(tmp_path))
    ref = transformers.BertTokenizer(str(tmp_path / "vocab.txt"), do_lower_case=False)
    for t in ("Hello hello É e", "HELLO"):
        assert mine.encode(t) == list(ref([t])["input_ids"][0])
