"""MI355X: the JETSGenerator protocol mirror and the CLI flow (BASELINE config 1 plumbing) end to end."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, rel_l2

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gen():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from emotivoice_amd.generator import JETSGeneratorHIP
    from emotivoice_amd.synthetic import synth_state_dict
    g = JETSGeneratorHIP(None).to("cuda:0")
    g.load_state_dict(synth_state_dict(0, "parity"))
    return g.eval()


def test_reference_call_pattern_b1_torch(gen):
    """Exactly the reference's call (inference_am_vocoder_joint.py:115-131) on the golden real-text line."""
    g = np.load(os.path.join(GOLDEN_DIR, "real_line1.npz"))
    device = "cuda:0"
    sequence = torch.from_numpy(g["in_ling"]).to(device).long().unsqueeze(0)
    sequence_len = torch.from_numpy(np.array([len(g["in_ling"])])).to(device)
    style = torch.from_numpy(g["in_style"]).to(device).unsqueeze(0)
    content = torch.from_numpy(g["in_content"]).to(device).unsqueeze(0)
    speaker = torch.from_numpy(np.array([int(g["in_speaker"])])).to(device)
    with torch.no_grad():
        out = gen(inputs_ling=sequence, inputs_style_embedding=style, input_lengths=sequence_len,
                  inputs_content_embedding=content, inputs_speaker=speaker, alpha=1.0)
    audio = out["wav_predictions"].squeeze() * 32768.0
    audio = audio.cpu().numpy().astype("int16")
    assert out["wav_predictions"].shape == (1, 1, 256 * int(g["mel_len"])) and out["wav_predictions"].is_cuda
    assert out["dec_outputs"].shape == (1, int(g["mel_len"]), 80)
    assert torch.equal(out["log_duration_predictions"].cpu(), torch.from_numpy(g["dur"]).unsqueeze(0))
    assert out["pitch_predictions"].shape == (len(g["in_ling"]),)
    assert rel_l2(out["dec_outputs"].cpu().numpy()[0], g["mel"]) < 1e-3
    assert rel_l2(out["wav_predictions"].cpu().numpy().ravel(), g["wav"]) < 1e-3
    assert audio.shape == (256 * int(g["mel_len"]),) and out["z_start_idxs"] is None and out["segment_size"] == 32
    for k in ("mel_targets", "postnet_outputs", "pitch_targets", "energy_targets", "duration_targets", "log_p_attn", "bin_loss"):
        assert out[k] is None


def test_padded_batch_numpy_inputs(gen):
    from emotivoice_amd.synthetic import synth_inputs
    utts = synth_inputs(61, [30, 12], [4, 9])
    ling = np.zeros((2, 30), np.int64)
    for b, u in enumerate(utts):
        ling[b, :len(u["ling"])] = u["ling"]
    out = gen(ling, np.array([30, 12]), np.array([4, 9]), np.stack([u["style"] for u in utts]), np.stack([u["content"] for u in utts]))
    solo = gen(utts[1]["ling"][None], np.array([12]), np.array([9]), utts[1]["style"][None], utts[1]["content"][None])
    T1 = solo["dec_outputs"].shape[1]
    assert np.array_equal(out["dec_outputs"][1, :T1], solo["dec_outputs"][0])      # B=1 semantics inside a padded batch
    assert np.all(out["dec_outputs"][1, T1:] == 0) and np.all(out["log_duration_predictions"][1, 12:] == 0)
    with pytest.raises(IndexError):
        gen(np.array([[600]]), np.array([1]), np.array([0]), utts[0]["style"][None], utts[0]["content"][None])
    with pytest.raises(NotImplementedError):
        gen(ling, np.array([30, 12]), np.array([4, 9]), np.zeros((2, 768), np.float32), np.zeros((2, 768), np.float32),
            mel_targets=np.zeros((2, 10, 80)))


def test_cli_flow_config1(tmp_path):
    """BASELINE configs[0] plumbing: text file -> ids -> generator -> int16 wav files (reference CLI flow)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import wave
    from emotivoice_amd import inference_am_vocoder_joint as cli
    toks = ["_", "<sos/eos>"] + ["p%d" % i for i in range(500)]
    (tmp_path / "tokenlist").write_text("\n".join(toks) + "\n")
    (tmp_path / "speaker2").write_text("\n".join(["8051"] + ["s%d" % i for i in range(2013)]))
    rng = np.random.default_rng(0)
    lines = []
    for n in (64, 9):
        ph = " ".join(["<sos/eos>"] + [toks[int(i)] for i in rng.integers(2, 502, n - 2)] + ["<sos/eos>"])
        lines.append("8051|Happy|%s|some content" % ph)
    lines.append("nobody|Happy|<sos/eos> p1 <sos/eos>|skipped: unknown speaker")
    (tmp_path / "text").write_text("\n".join(lines) + "\n")
    n = cli.main(["-t", str(tmp_path / "text"), "--tokenlist", str(tmp_path / "tokenlist"), "--speakers", str(tmp_path / "speaker2"),
                  "--synthetic-weights", "-o", str(tmp_path / "out")])
    assert n == 2 and sorted(os.listdir(tmp_path / "out")) == ["1.wav", "2.wav"]
    with wave.open(str(tmp_path / "out" / "1.wav")) as w:
        assert w.getframerate() == 16000 and w.getsampwidth() == 2 and w.getnframes() % 256 == 0 and w.getnframes() > 0


def test_predictor_tts_surface(tmp_path):
    """predict.py:164-194 flow through emotivoice_amd.predict.Predictor."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import wave
    from emotivoice_amd.predict import Predictor
    from emotivoice_amd.synthetic import synth_state_dict
    toks = ["_", "<sos/eos>"] + ["p%d" % i for i in range(500)]
    (tmp_path / "tokenlist").write_text("\n".join(toks) + "\n")
    (tmp_path / "speaker2").write_text("\n".join(["8051"] + ["s%d" % i for i in range(2013)]))
    p = Predictor(str(tmp_path / "tokenlist"), str(tmp_path / "speaker2"), str(tmp_path / "out"))
    with pytest.raises(RuntimeError):
        p.tts("<sos/eos> p1 <sos/eos>", "Happy", "hello", "8051")
    p.setup_models(generator_state_dict=synth_state_dict(0, "parity"))
    path = p.tts("<sos/eos> p1 p2 p3 p4 p5 <sos/eos>", "Happy", "hello there", "8051")
    with wave.open(path) as w:
        assert w.getframerate() == 16000 and w.getnframes() % 256 == 0 and w.getnframes() > 0
    with pytest.raises(KeyError):
        p.tts("<sos/eos> nope <sos/eos>", "Happy", "x", "8051")
    with pytest.raises(KeyError):
        p.tts("<sos/eos> p1 <sos/eos>", "Happy", "x", "unknown-speaker")
