"""A G2P of realistic cost for front-end throughput measurements.

The reference's English branch (frontend_en.py:38-78) is a regex split into words and separators, one lower-cased dictionary lookup
per word in the 135 k-entry librispeech lexicon, a per-phone loop that brackets ARPAbet symbols, ``engsp1`` between words, ``engsp4`` at
punctuation, and g2p_en's neural model for out-of-lexicon words.  Neither the lexicon file nor g2p_en / jieba / pypinyin are in this
image, so this module reproduces the WORK with stand-in data: a seeded synthetic lexicon of the same size and phone-count distribution
(1-12 ARPAbet-like phones per word), the same per-line control flow written from the description above (not from the reference's
source), and a cheap letter-to-sound rule for unknown words (the neural fallback is far slower, but rare on lexicon text).  It is used
only by tools/bench_frontend.py and bench.py --mode pipeline; the product takes the caller's real G2P (frontend_pool.FrontendPool)."""
from __future__ import annotations

import re
from typing import Dict, List

import numpy as np

PHONES = ["AA0", "AA1", "AE1", "AH0", "AH1", "AO1", "AW1", "AY1", "B", "CH", "D", "DH", "EH1", "ER0", "EY1", "F", "G", "HH", "IH0", "IH1", "IY0",
          "IY1", "JH", "K", "L", "M", "N", "NG", "OW1", "OY1", "P", "R", "S", "SH", "T", "TH", "UH1", "UW1", "V", "W", "Y", "Z", "ZH"]
_SPLIT = re.compile(r"([,;.\-\?\!\s+])")
_LETTERS = "abcdefghijklmnopqrstuvwxyz"


def make_lexicon(n_words: int = 135000, seed: int = 0) -> Dict[str, List[str]]:
    rng = np.random.default_rng(seed)
    lex: Dict[str, List[str]] = {}
    lens = rng.integers(2, 12, size=n_words)
    nph = np.clip(rng.poisson(5.5, size=n_words), 1, 12)
    letters = rng.integers(0, 26, size=int(lens.sum()))
    phones = rng.integers(0, len(PHONES), size=int(nph.sum()))
    a = b = 0
    for i in range(n_words):
        w = "".join(_LETTERS[j] for j in letters[a:a + lens[i]])
        a += lens[i]
        if w not in lex:
            lex[w] = [PHONES[j] for j in phones[b:b + nph[i]]]
        b += nph[i]
    return lex


def make_text(lex: Dict[str, List[str]], n_lines: int, words_per_line: int = 14, seed: int = 1, oov_rate: float = 0.01) -> List[str]:
    """Lines of Zipf-distributed lexicon words with commas / full stops, a few out-of-lexicon words."""
    rng = np.random.default_rng(seed)
    vocab = list(lex.keys())
    ranks = np.minimum(rng.zipf(1.3, size=n_lines * words_per_line) - 1, len(vocab) - 1)
    out = []
    k = 0
    for _ in range(n_lines):
        ws = []
        for j in range(words_per_line):
            w = vocab[ranks[k]]
            k += 1
            if rng.random() < oov_rate:
                w = w + "zq"
            if j % 5 == 4 and j + 1 < words_per_line:
                w += ","
            ws.append(w.capitalize() if j == 0 else w)
        out.append(" ".join(ws) + ".")
    return out


def _rule_g2p(word: str) -> List[str]:
    return [PHONES[(ord(c) * 7 + i) % len(PHONES)] for i, c in enumerate(word) if c.isalpha()]


def make_g2p(lex: Dict[str, List[str]]):
    skip = {",", " ", "'"}

    def g2p(text: str) -> str:
        phones: List[str] = []
        for w in (t for t in _SPLIT.split(text) if t not in ("", " ")):
            entry = lex.get(w.lower())
            if entry is not None:
                for ph in entry:
                    if ph not in skip:
                        phones.append("[" + ph + "]")
                if "sp" not in phones[-1]:
                    phones.append("engsp1")
            elif w.isalpha():
                for ph in _rule_g2p(w.lower()):
                    phones.append("[" + ph + "]")
                phones.append("engsp1")
            elif phones and not w.isspace():
                phones.pop()
                phones.append("engsp4")
        if phones and "engsp" in phones[-1]:
            phones.pop()
        return " ".join(["<sos/eos>"] + phones + ["<sos/eos>"])
    return g2p


def token_table() -> Dict[str, int]:
    toks = ["<sos/eos>", "engsp1", "engsp4"] + ["[" + p + "]" for p in PHONES]
    return {t: i for i, t in enumerate(toks)}
