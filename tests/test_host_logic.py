"""CPU tests of the host-side mirror of the reference interface: config mapping, text-line contract,
int16 epilogue, utterance sharding, generator protocol error behaviour (no GPU needed)."""
import os

import numpy as np
import pytest

from emotivoice_amd.config import EVShapes, from_reference_config, load_yaml
from emotivoice_amd.sharding import shard_utterances
from emotivoice_amd.text_io import (HashStyleEmbedder, parse_line, phonemes_to_ids, read_table, wav_float_to_int16,
                                    write_wav_int16)

REF_YAML = """
sr: 16000
hop_length: 256
n_mels: 80
segment_size: 32
model:
    bert_embedding: 768
    encoder_n_layers: 4
    encoder_n_heads: 8
    encoder_n_hidden: 384
    encoder_kernel_size_conv_mod: 3
    decoder_n_layers: 4
    decoder_n_heads: 8
    decoder_n_hidden: 384
    variance_n_hidden: 384
    variance_n_layers: 3
    variance_kernel_size: 3
    variance_embed_kernel_size: 9
    duration_n_layers: 2
    resblock: "1"
    upsample_rates: [8,8,2,2]
    upsample_kernel_sizes: [16,16,4,4]
    upsample_initial_channel: 512
    resblock_kernel_sizes: [3,7,11]
    resblock_dilation_sizes: [[1,3,5], [1,3,5], [1,3,5]]
"""


def test_reference_config_maps_to_defaults(tmp_path):
    p = tmp_path / "config.yaml"
    p.write_text(REF_YAML)
    conf = load_yaml(str(p))
    assert conf.n_vocab == 502 and conf.n_speaker == 2014 and conf.model.encoder_n_hidden == 384
    assert from_reference_config(conf) == EVShapes()
    assert from_reference_config(None) == EVShapes() and EVShapes().upsample_factor == 256


def test_reference_config_rejects_unsupported(tmp_path):
    p = tmp_path / "config.yaml"
    p.write_text(REF_YAML.replace('resblock: "1"', 'resblock: "2"'))
    with pytest.raises(ValueError):
        from_reference_config(load_yaml(str(p)))
    p.write_text(REF_YAML.replace("hop_length: 256", "hop_length: 300"))
    with pytest.raises(ValueError):
        from_reference_config(load_yaml(str(p)))


def test_text_line_contract(tmp_path):
    # README example shape: speaker|prompt|phonemes|content
    line = "8051|Happy|<sos/eos> [IH0] [M] [AA1] [T] engsp4 [V] [OY1] [S] <sos/eos>|Emoti-Voice"
    ln = parse_line(line)
    assert ln.speaker == "8051" and ln.prompt == "Happy" and ln.content == "Emoti-Voice"
    assert ln.phonemes[0] == "<sos/eos>" and len(ln.phonemes) == 10
    tl = tmp_path / "tokenlist"
    toks = ["_", "<sos/eos>", "[IH0]", "[M]", "[AA1]", "[T]", "engsp4", "[V]", "[OY1]", "[S]"]
    tl.write_text("\n".join(toks))        # unterminated last line, like the reference's speaker2
    t2i = read_table(str(tl))
    assert t2i["_"] == 0 and t2i["<sos/eos>"] == 1 and t2i["[S]"] == 9
    ids = phonemes_to_ids(ln.phonemes, t2i)
    assert ids.dtype == np.int64 and ids.tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 9, 1]
    with pytest.raises(KeyError):
        phonemes_to_ids(["[ZZ9]"], t2i)
    with pytest.raises(ValueError):
        parse_line("only|three|fields")


def test_style_embedder_is_deterministic_and_bounded():
    e = HashStyleEmbedder()
    a, b = e("Happy"), e("Happy")
    assert a.shape == (768,) and a.dtype == np.float32 and np.array_equal(a, b)
    assert np.abs(a).max() < 1.0 and not np.array_equal(a, e("Sad"))


def test_int16_epilogue_and_wav_file(tmp_path):
    x = np.array([0.0, 0.5, -0.5, 0.99999, -1.0, 1.5 / 32768, -1.5 / 32768], np.float32)
    assert wav_float_to_int16(x).tolist() == [0, 16384, -16384, 32767, -32768, 1, -1]
    p = tmp_path / "a.wav"
    write_wav_int16(str(p), wav_float_to_int16(x))
    import wave
    with wave.open(str(p)) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 16000, 7)


def test_shard_utterances_balances_and_partitions():
    lens = [64 + (i * 7919) % 449 for i in range(256)]      # BASELINE config 3 length spread
    shards = shard_utterances(lens, 8)
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(256))
    loads = [sum(lens[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(lens)
    assert shard_utterances([5, 5, 5], 1) == [[0, 1, 2]]
    assert shard_utterances(lens, 8) == shards             # deterministic


def test_generator_default_precision_is_the_contract_mode():
    """The one-line swap of INTEGRATION.md (no precision argument) must land in the mode that meets the 1e-3 waveform contract; an explicit
    component precision keeps the low-level resolution rule."""
    from emotivoice_amd.engine import resolve_precision
    from emotivoice_amd.generator import JETSGeneratorHIP
    g = JETSGeneratorHIP(None)
    assert resolve_precision(g._precision, g._dec_prec, g._voc_prec) == ("mx", "mx")
    g = JETSGeneratorHIP(None, precision="fast")
    assert resolve_precision(g._precision, g._dec_prec, g._voc_prec) == ("f16", "f16")
    g = JETSGeneratorHIP(None, vocoder_precision="x3")
    assert resolve_precision(g._precision, g._dec_prec, g._voc_prec) == ("mx", "x3")      # the unnamed component stays in the contract mode
    assert resolve_precision(None, None, None) == ("mx", "mx")            # EVEngine() and ev_default_config (ABI 5) agree with the drop-in object
    with pytest.raises(ValueError):
        resolve_precision("fp8", None, None)


def test_generator_protocol_errors_without_gpu():
    import torch
    from emotivoice_amd.engine import EVError
    from emotivoice_amd.generator import JETSGeneratorHIP
    g = JETSGeneratorHIP(None)
    assert g.eval() is g and g.segment_size == 32 and g.upsample_factor == 256
    with pytest.raises(NotImplementedError):
        g.train(True)
    with pytest.raises(EVError):
        g.to("cpu")
    with pytest.raises(RuntimeError, match="Unexpected key"):
        g.load_state_dict({"bogus.weight": np.zeros(1)})
    with pytest.raises(KeyError):
        g.load_state_dict({"am.to_mel.weight": np.zeros((80, 384), np.float32)})
    if not torch.cuda.is_available():
        with pytest.raises(EVError):
            g.to("cuda:0")


def test_g2p_standin_is_deterministic_and_pool_consistent():
    """The front-end stand-in used by tools/bench_frontend.py and bench.py --mode pipeline (the reference's English lexicon-path work on a
    seeded synthetic lexicon): deterministic, every output token is in its token table, and FrontendPool returns exactly the serial result."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from g2p_standin import make_g2p, make_lexicon, make_text, token_table
    from emotivoice_amd.frontend_pool import FrontendPool
    lex = make_lexicon(5000, seed=3)
    assert lex == make_lexicon(5000, seed=3) and len(lex) > 4000
    g2p = make_g2p(lex)
    texts = make_text(lex, 200, words_per_line=9, seed=4, oov_rate=0.05)
    out = [g2p(t) for t in texts]
    tab = token_table()
    assert all(tok in tab for line in out for tok in line.split())
    assert all(line.startswith("<sos/eos> ") and line.endswith(" <sos/eos>") and "engsp" not in line.split()[-2] for line in out)
    assert any("engsp4" in line for line in out) and any("engsp1" in line for line in out)
    with FrontendPool(g2p, workers=2, chunk=16) as pool:
        assert pool.map(texts) == out
