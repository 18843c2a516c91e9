"""BERT WordPiece tokenizer for the SimBERT prompt / content encoder, from ``vocab.txt`` alone.

The reference tokenises prompts with ``AutoTokenizer.from_pretrained(config.bert_path)`` (inference_am_vocoder_joint.py:83,
inference_tts.py:84, predict.py:120) and calls it as ``tokenizer([text], return_tensors="pt")`` -> ``input_ids``, ``token_type_ids``,
``attention_mask`` (:26-29).  For a BERT checkpoint such as simbert-base-chinese that is the classic two-stage algorithm, restated here so
that the device path has no dependency on ``transformers`` (and no hub lookups): basic tokenisation -- control characters dropped, whitespace
normalised, every CJK ideograph split off as its own token, optional lower-casing with accent stripping, punctuation split off -- then greedy
longest-match-first WordPiece with ``##`` continuation pieces, ``[UNK]`` for words without a segmentation or longer than 100 characters,
wrapped in ``[CLS] ... [SEP]``.  ``tests/test_wordpiece.py`` compares it with ``transformers.BertTokenizer`` on mixed Chinese / English /
punctuation / accent / unknown-word inputs over a vocabulary written by the test."""
from __future__ import annotations

import os
import unicodedata
from typing import Dict, List, Optional, Sequence

import numpy as np


def _is_whitespace(ch: str) -> bool:
    return ch in (" ", "\t", "\n", "\r") or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in ("\t", "\n", "\r"):
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    # all non-letter / non-digit ASCII counts as punctuation ("^", "$", "`" are not in the Unicode P* classes)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    def __init__(self, vocab_path: str, do_lower_case: bool = True, unk_token: str = "[UNK]", cls_token: str = "[CLS]",
                 sep_token: str = "[SEP]", pad_token: str = "[PAD]", max_input_chars_per_word: int = 100):
        if os.path.isdir(vocab_path):
            vocab_path = os.path.join(vocab_path, "vocab.txt")
        with open(vocab_path, "r", encoding="utf-8") as f:
            self.vocab: Dict[str, int] = {line.rstrip("\n"): i for i, line in enumerate(f)}
        self.do_lower_case = do_lower_case
        self.unk_token, self.cls_token, self.sep_token, self.pad_token = unk_token, cls_token, sep_token, pad_token
        self.special = {unk_token, cls_token, sep_token, pad_token, "[MASK]"}
        self.max_chars = max_input_chars_per_word
        for t in (unk_token, cls_token, sep_token, pad_token):
            if t not in self.vocab:
                raise ValueError("vocabulary has no %s" % t)

    # ---- stage 1
    def _clean(self, text: str) -> str:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            out.append(" " if _is_whitespace(ch) else ch)
        return "".join(out)

    def basic_tokenize(self, text: str) -> List[str]:
        text = self._clean(text)
        text = "".join(" %s " % ch if _is_cjk(ord(ch)) else ch for ch in text)
        text = unicodedata.normalize("NFC", text)
        tokens: List[str] = []
        for tok in text.split():
            if tok in self.special:                      # (never_split: the special tokens pass through untouched)
                tokens.append(tok)
                continue
            if self.do_lower_case:
                tok = tok.lower()
                tok = "".join(ch for ch in unicodedata.normalize("NFD", tok) if unicodedata.category(ch) != "Mn")
            cur: List[str] = []
            for ch in tok:                               # split punctuation off
                if _is_punctuation(ch):
                    if cur:
                        tokens.append("".join(cur))
                        cur = []
                    tokens.append(ch)
                else:
                    cur.append(ch)
            if cur:
                tokens.append("".join(cur))
        return tokens

    # ---- stage 2
    def wordpiece(self, word: str) -> List[str]:
        if len(word) > self.max_chars:
            return [self.unk_token]
        pieces, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = word[start:end] if start == 0 else "##" + word[start:end]
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk_token]
            pieces.append(cur)
            start = end
        return pieces

    def tokenize(self, text: str) -> List[str]:
        out: List[str] = []
        for tok in self.basic_tokenize(text):
            out.extend([tok] if tok in self.special else self.wordpiece(tok))
        return out

    def encode(self, text: str, max_length: Optional[int] = None) -> List[int]:
        toks = self.tokenize(text)
        if max_length is not None:
            toks = toks[:max(0, max_length - 2)]
        return [self.vocab[self.cls_token]] + [self.vocab[t] for t in toks] + [self.vocab[self.sep_token]]

    def __call__(self, texts, return_tensors: Optional[str] = "np", max_length: Optional[int] = 512):
        """``tokenizer([text, ...])`` -> dict of right-padded int64 arrays (numpy; ``return_tensors='pt'`` gives torch tensors)."""
        if isinstance(texts, str):
            texts = [texts]
        ids = [self.encode(t, max_length) for t in texts]
        n = max(len(x) for x in ids)
        pad = self.vocab[self.pad_token]
        input_ids = np.full((len(ids), n), pad, np.int64)
        mask = np.zeros((len(ids), n), np.int64)
        for b, x in enumerate(ids):
            input_ids[b, :len(x)] = x
            mask[b, :len(x)] = 1
        out = dict(input_ids=input_ids, token_type_ids=np.zeros_like(input_ids), attention_mask=mask)
        if return_tensors == "pt":
            import torch
            out = {k: torch.from_numpy(v) for k, v in out.items()}
        return out


def load_tokenizer(bert_path: str):
    """The tokenizer of ``config.bert_path``: this module's WordPiece when the directory holds a ``vocab.txt`` (lower-casing read from
    ``tokenizer_config.json`` when present, default True like bert-base-chinese), else whatever ``transformers.AutoTokenizer`` finds there."""
    vocab = os.path.join(bert_path, "vocab.txt")
    if os.path.exists(vocab):
        lower = True
        cfg = os.path.join(bert_path, "tokenizer_config.json")
        if os.path.exists(cfg):
            import json
            with open(cfg, "r", encoding="utf-8") as f:
                lower = bool(json.load(f).get("do_lower_case", True))
        return WordPieceTokenizer(vocab, do_lower_case=lower)
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(bert_path, local_files_only=True)
