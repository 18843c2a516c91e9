"""The reference's text-line input contract: ``<speaker>|<prompt>|<phoneme tokens>|<content>``
(inference_am_vocoder_joint.py:96-102, README.md:97-99), token -> id by line index of ``tokenlist`` and
speaker -> id by line index of ``speaker2`` (config/joint/config.py:56-60, inference_am_vocoder_joint.py:76-81)."""
from __future__ import annotations

import hashlib
import wave
from typing import Dict, List, NamedTuple

import numpy as np


class InputLine(NamedTuple):
    speaker: str
    prompt: str
    phonemes: List[str]
    content: str


def read_table(path: str) -> Dict[str, int]:
    """token / speaker -> id by line index, keys stripped like the reference's ``t.strip()`` (inference_am_vocoder_joint.py:76-80;
    the speaker file is read as utf-8 there, and the token file is ASCII)."""
    with open(path, "r", encoding="utf-8") as f:
        return {line.strip(): i for i, line in enumerate(f.readlines())}


def parse_line(line: str) -> InputLine:
    parts = line.strip().split("|")
    if len(parts) < 4:
        raise ValueError("expected <speaker>|<prompt>|<phoneme>|<content>, got %r" % line)
    return InputLine(parts[0], parts[1], parts[2].split(), parts[3])


def read_text_file(path: str) -> List[InputLine]:
    with open(path, "r", encoding="utf-8") as f:
        return [parse_line(l) for l in f if l.strip()]


def phonemes_to_ids(phonemes: List[str], token2id: Dict[str, int]) -> np.ndarray:
    # unknown phoneme -> KeyError, like the reference's ``[token2id[ph] for ph in text]``
    return np.array([token2id[ph] for ph in phonemes], np.int64)


class HashStyleEmbedder:
    """Placeholder for the SimBERT prompt/content encoder (models/prompt_tts_modified/simbert.py:48-72), which is out of
    scope for the hot path (HF weights are not available offline).  Produces a deterministic
    768-d vector in (-1, 1) -- the range of a BERT pooler output -- from the text, so the CLI plumbing can be exercised.
    Real deployments pass embeddings computed by the reference's StyleEncoder (or its cached .npy files)."""

    def __init__(self, dim: int = 768):
        self.dim = dim

    def __call__(self, text: str) -> np.ndarray:
        seed = int.from_bytes(hashlib.sha256(text.encode("utf-8")).digest()[:8], "little")
        return np.tanh(np.random.default_rng(seed).standard_normal(self.dim)).astype(np.float32)


def wav_float_to_int16(wav: np.ndarray, max_wav_value: float = 32768.0) -> np.ndarray:
    """inference_am_vocoder_joint.py:130-131: ``(wav * MAX_WAV_VALUE).astype('int16')`` (C cast: truncation, wrap)."""
    return (np.asarray(wav, np.float32) * np.float32(max_wav_value)).astype(np.int64).astype(np.int16)


def write_wav_int16(path: str, audio_i16: np.ndarray, sample_rate: int = 16000) -> None:
    """16-bit PCM mono (soundfile, used by the reference at :134, is not needed for plain PCM)."""
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.ascontiguousarray(audio_i16, np.int16).tobytes())
