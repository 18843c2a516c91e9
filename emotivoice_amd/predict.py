"""MI355X counterpart of the hot-path part of the reference's cog ``Predictor`` (predict.py:99-194).

Kept: ``setup_models`` (load generator + vocab tables, predict.py:101-136) and ``tts(text, prompt, content, speaker)``
(predict.py:164-194: phoneme string -> ids, one generator call with B = 1, ``wav * 32768 -> int16``, write the file).
Not here (out of the hot-path scope): weight download, the SimBERT style encoder
(``style_embedder`` is pluggable; the default is the documented placeholder of text_io.HashStyleEmbedder), the G2P
front-end of ``predict`` (pass the phoneme string the reference's frontend produces) and mp3 encoding (16-bit PCM wav).
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import numpy as np

from .generator import JETSGeneratorHIP
from .text_io import HashStyleEmbedder, read_table, wav_float_to_int16, write_wav_int16

MAX_WAV_VALUE = 32768.0          # models/hifigan/get_vocoder.py (imported by the reference callers)


class Predictor:
    def __init__(self, token_list_path: str, speaker2id_path: str, output_directory: str = ".", conf=None,
                 device: str = "cuda:0", style_embedder: Optional[Callable[[str], np.ndarray]] = None,
                 g2p: Optional[dict] = None, precision: str = "mx"):
        """``g2p``: {"English": fn(text) -> phoneme string, "Chinese": fn} -- the reference's frontend_en.g2p_en / frontend_cn.g2p_cn
        (jieba / pypinyin / g2p_en are host-side third-party packages outside the hot path; pass them in when installed)."""
        self.conf = conf
        self.device = device
        self.output_directory = output_directory
        self.token2id = read_table(token_list_path)       # predict.py:126-127
        self.speaker2id = read_table(speaker2id_path)     # predict.py:129-130
        self.style_embedder = style_embedder or HashStyleEmbedder()
        self.g2p = g2p or {}
        self.precision = precision
        self.generator: Optional[JETSGeneratorHIP] = None
        self.style_encoder = None
        self.tokenizer = None
        self.sampling_rate = 16000

    def setup_models(self, generator_state_dict=None, checkpoint_path: Optional[str] = None, style_encoder_state_dict=None,
                     tokenizer=None):
        """predict.py:101-136: JETSGenerator(conf).to(device); load_state_dict(ckpt['generator']); eval() -- and, when a
        StyleEncoder state dict and a tokenizer are given, the SimBERT encoder on the device (predict.py:109-117,124)."""
        if generator_state_dict is None:
            if checkpoint_path is None:
                raise ValueError("generator_state_dict or checkpoint_path is required")
            import torch
            generator_state_dict = torch.load(checkpoint_path, map_location="cpu")["generator"]
        gen = JETSGeneratorHIP(self.conf, precision=self.precision).to(self.device)
        gen.load_state_dict(generator_state_dict)
        self.generator = gen.eval()
        self.sampling_rate = gen.shapes.sr
        if style_encoder_state_dict is not None:
            from .simbert import StyleEncoderHIP
            enc = StyleEncoderHIP(None, device=self.device, engine=gen._ensure_engine())      # one handle, one GPU context
            enc.load_state_dict(style_encoder_state_dict, strict=False)
            self.style_encoder, self.tokenizer = enc.eval(), tokenizer
        return self

    def get_style_embedding(self, prompt: str) -> np.ndarray:
        """predict.py:142-158: tokenizer([text]) -> StyleEncoder -> pooled_output.  Without a loaded style encoder the pluggable
        ``style_embedder`` (default: the documented placeholder) is used."""
        if self.style_encoder is not None and self.tokenizer is not None:
            t = self.tokenizer([prompt], return_tensors="np")
            out = self.style_encoder(input_ids=t["input_ids"], token_type_ids=t["token_type_ids"], attention_mask=t["attention_mask"])
            return np.asarray(out["pooled_output"], np.float32).squeeze()
        return np.asarray(self.style_embedder(prompt), np.float32)

    def tts(self, text: str, prompt: str, content: str, speaker: str, filename: str = "output.wav") -> str:
        """text: space-separated phoneme tokens (what the reference's g2p front-end returns).  KeyError for an unknown
        speaker or phoneme, exactly like predict.py:169-171."""
        if self.generator is None:
            raise RuntimeError("setup_models() first")
        style = self.get_style_embedding(prompt)
        content_emb = self.get_style_embedding(content)
        spk = self.speaker2id[speaker]
        text_int = np.array([self.token2id[ph] for ph in text.split()], np.int64)
        out = self.generator(inputs_ling=text_int[None], inputs_style_embedding=style[None],
                             input_lengths=np.array([len(text_int)]), inputs_content_embedding=content_emb[None],
                             inputs_speaker=np.array([spk]), alpha=1.0)
        audio = wav_float_to_int16(np.asarray(out["wav_predictions"]).squeeze(), MAX_WAV_VALUE)
        os.makedirs(self.output_directory, exist_ok=True)
        path = os.path.join(self.output_directory, filename)
        write_wav_int16(path, audio, self.sampling_rate)
        return path

    def predict(self, prompt: str = "Happy", content: str = "Emoti-Voice - a Multi-Voice and Prompt-Controlled T-T-S Engine",
                language: str = "English", speaker: Optional[str] = None) -> str:
        """predict.py:196-234: language check, G2P, then tts().  Same ValueErrors for a language / script mismatch."""
        if speaker is None:
            speaker = next(iter(self.speaker2id))
        if language not in ("English", "Chinese"):
            raise ValueError("language must be 'English' or 'Chinese'")
        has_cn = contains_chinese(content)
        if language == "English" and has_cn:
            raise ValueError("文本含有中文/input text contains Chinese, but language is English")
        if language == "Chinese" and not has_cn:
            raise ValueError("文本含有英文/input text contains English, but language is Chinese")
        if language not in self.g2p:
            raise RuntimeError("no G2P front-end for %s: pass g2p={'%s': fn} (the reference uses frontend_%s.py)" %
                               (language, language, "en" if language == "English" else "cn"))
        text = self.g2p[language](content)
        return self.tts(text, prompt, content, speaker)


def contains_chinese(text: str) -> bool:
    """frontend.py:61-64."""
    import re
    return re.search(r"[\u4e00-\u9fa5]", text) is not None
