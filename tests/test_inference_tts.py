"""Bulk driver mirror (emotivoice_amd/inference_tts.py) against the behaviour of the reference's inference_tts.py:89-156,197-222:
chunk arithmetic, prompt / speaker cycling, file layout, resume-by-existing-file, per-line error isolation.  Host logic only: the
engine is a stub that returns one ramp per utterance."""
import os
import wave

import numpy as np
import pytest

from emotivoice_amd.inference_tts import PROMPTS, build_parser, run_chunk, split_chunks, utt_paths


def test_split_chunks_matches_reference_arithmetic():
    # total >= workers: floor division, the first (total % n) workers one longer (:197-222)
    assert split_chunks(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert split_chunks(9, 3) == [(0, 3), (3, 3), (6, 3)]
    assert split_chunks(2048, 8) == [(256 * j, 256) for j in range(8)]
    # fewer lines than workers: chunk size 1, the surplus workers get ranges past the end
    assert split_chunks(2, 4) == [(0, 1), (1, 1), (2, 1), (3, 1)]
    for total, n in [(1, 1), (7, 2), (100, 16), (5, 8)]:
        parts = split_chunks(total, n)
        covered = [i for b, c in parts for i in range(b, b + c) if i < total]
        assert covered == list(range(total))                    # contiguous, in order, nothing twice


def test_parser_has_the_reference_flags_and_defaults():
    a = build_parser().parse_args(["-t", "/x/texts.txt"])
    assert (a.logdir, a.config_folder, a.checkpoint, a.output_dir, a.gpu_ids, a.num_thread) == \
        ("prompt_tts_open_source_joint", "config/joint", "g_00140000", None, "0", "1")


class _StubEngine:
    def __init__(self):
        self.calls = []

    def synthesize(self, utts):
        self.calls.append(utts)
        return dict(wav_list=[np.linspace(-0.5, 0.5, 10 * len(u["ling"]), dtype=np.float32) for u in utts])


def _embed(text):
    return np.full(768, (sum(text.encode("utf-8")) % 97) / 97.0, np.float32)


def test_run_chunk_layout_cycling_resume_and_errors(tmp_path):
    token2id = {t: i for i, t in enumerate(["a", "b", "c", "sp"])}
    id2speaker = {0: "spk_zero", 1: "spk_one", 2: "spk_two"}
    lines = ["a b c\n", "b sp a\n", "a X c\n", "c c\n", "a\n", "b b b b\n", "sp\n"]        # line 2 has an unknown phoneme
    out = str(tmp_path / "audio")
    # line 3 was synthesised by an earlier run: must be skipped before any work is done for it
    d3, wav3, _ = utt_paths(out, id2speaker[3 % 3], 3)
    os.makedirs(d3)
    open(wav3, "wb").write(b"old")
    eng, logs = _StubEngine(), []
    stats = run_chunk(lines, 1, 5, synthesize=eng.synthesize, embed=_embed, g2p=lambda s: s, token2id=token2id, id2speaker=id2speaker,
                      output_dir=out, sampling_rate=16000, batch=2, log=logs.append)
    assert stats == dict(written=3, skipped_existing=1, errors=1)
    assert open(wav3, "rb").read() == b"old"
    assert any("exists, continue" in m for m in logs) and any(m.startswith("Error:") for m in logs)
    # lines 1, 4, 5 were written (0 and 6 are outside the chunk, 2 failed, 3 existed): <speaker name>/<i+1:06d>.wav + .txt
    for i in (1, 4, 5):
        d, w, t = utt_paths(out, id2speaker[i % 3], i)
        assert os.path.basename(w) == "%06d.wav" % (i + 1)
        with wave.open(w, "rb") as f:
            assert (f.getframerate(), f.getnchannels(), f.getsampwidth()) == (16000, 1, 2)
            pcm = np.frombuffer(f.readframes(f.getnframes()), np.int16)
        n = len(lines[i].split())
        expect = (np.linspace(-0.5, 0.5, 10 * n, dtype=np.float32) * np.float32(32768.0)).astype(np.int64).astype(np.int16)
        assert np.array_equal(pcm, expect)
        assert open(t, encoding="utf-8").read() == lines[i].strip() + "\n"
    for i in (0, 2, 6):
        assert not os.path.exists(utt_paths(out, id2speaker[i % 3], i)[1])
    # the engine saw batches of <= 2 utterances with prompt PROMPTS[i % 4], speaker i % n_speaker, content embedding of the line
    seen = [u for call in eng.calls for u in call]
    assert [len(c) for c in eng.calls] == [1, 2]              # group (1, 2): line 2 dropped; group (4, 5)
    for u, i in zip(seen, (1, 4, 5)):
        assert u["speaker"] == i % 3
        assert np.array_equal(u["style"], _embed(PROMPTS[i % 4]))
        assert np.array_equal(u["content"], _embed(lines[i].strip()))
        assert list(u["ling"]) == [token2id[t] for t in lines[i].split()]


def test_run_chunk_failing_batch_does_not_stop_the_run(tmp_path):
    token2id, id2speaker = {"a": 0}, {0: "s"}

    def synth(utts):
        if len(utts[0]["ling"]) == 2:
            raise RuntimeError("device said no")
        return dict(wav_list=[np.zeros(4, np.float32) for _ in utts])
    stats = run_chunk(["a\n", "a a\n", "a a a\n"], 0, 3, synthesize=synth, embed=_embed, g2p=lambda s: s, token2id=token2id,
                      id2speaker=id2speaker, output_dir=str(tmp_path), sampling_rate=16000, batch=1, log=lambda m: None)
    assert stats == dict(written=2, skipped_existing=0, errors=1)


def test_run_chunk_bad_line_costs_one_line_not_its_batch(tmp_path):
    """with --batch 3 a line the engine rejects used to drop all three; the batch is re-run line by line."""
    calls = []

    def synth(utts):
        calls.append(len(utts))
        if any(len(u["ling"]) == 2 for u in utts):
            raise RuntimeError("device said no")
        return dict(wav_list=[np.zeros(4, np.float32) for _ in utts])
    stats = run_chunk(["a\n", "a a\n", "a a a\n"], 0, 3, synthesize=synth, embed=_embed, g2p=lambda s: s, token2id={"a": 0},
                      id2speaker={0: "s"}, output_dir=str(tmp_path), sampling_rate=16000, batch=3, log=lambda m: None)
    assert stats == dict(written=2, skipped_existing=0, errors=1) and calls == [3, 1, 1, 1]
    assert os.path.exists(utt_paths(str(tmp_path), "s", 0)[1]) and os.path.exists(utt_paths(str(tmp_path), "s", 2)[1])
    assert not os.path.exists(utt_paths(str(tmp_path), "s", 1)[1])


def test_g2p_map_is_used_when_given(tmp_path):
    token2id, id2speaker = {"x": 0, "y": 1}, {0: "s"}
    eng = _StubEngine()
    run_chunk(["hello\n", "world\n"], 0, 2, synthesize=eng.synthesize, embed=_embed, g2p=lambda s: pytest.fail("per-line G2P used"),
              token2id=token2id, id2speaker=id2speaker, output_dir=str(tmp_path), sampling_rate=16000, batch=8,
              g2p_map=lambda texts: ["x y" if t == "hello" else "y" for t in texts], log=lambda m: None)
    assert [list(u["ling"]) for u in eng.calls[0]] == [[0, 1], [1]]
