"""Text front-end throughput.

The reference's G2P (frontend.py:23-59, frontend_cn.py:102-121, frontend_en.py:38-78) is pure-Python jieba / pypinyin / g2p_en
running one line at a time in the synthesis process (inference_tts.py:63-71).  With the generator at >10^4 utterances per second
per node that loop is the next host bottleneck; it is embarrassingly parallel over lines and touches no GPU.  ``FrontendPool`` fans
lines out to worker processes (fork: the G2P's lexicons and models are loaded once and shared copy-on-write) in chunks, keeps
the input order and bounds the number of chunks in flight so that it can sit in front of a DynamicBatcher as a stream.
The G2P callable itself is supplied by the caller (the reference's ``g2p_cn_en`` with its lexicon / G2p objects bound).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import time
from typing import Callable, Iterable, Iterator, List, Optional, Sequence

_G2P: Optional[Callable[[str], str]] = None


def _init(fn):
    global _G2P
    _G2P = fn


def _work(chunk: Sequence[str]) -> List[str]:
    return [_G2P(t) for t in chunk]


class FrontendPool:
    def __init__(self, g2p: Callable[[str], str], workers: Optional[int] = None, chunk: int = 64):
        self.g2p, self.chunk = g2p, max(1, chunk)
        self.workers = workers or max(1, min(32, (os.cpu_count() or 2) - 1))
        self._pool = None
        if self.workers > 1:
            ctx = mp.get_context("fork")
            self._pool = ctx.Pool(self.workers, initializer=_init, initargs=(g2p,))

    def map(self, texts: Sequence[str]) -> List[str]:
        """Phoneme strings of ``texts`` in input order."""
        if self._pool is None:
            return [self.g2p(t) for t in texts]
        chunks = [texts[i:i + self.chunk] for i in range(0, len(texts), self.chunk)]
        out: List[str] = []
        for part in self._pool.imap(_work, chunks):          # imap keeps order; results stream back as chunks finish
            out.extend(part)
        return out

    def stream(self, texts: Iterable[str]) -> Iterator[str]:
        """Lazily consume an iterable of lines, yielding phoneme strings in order with bounded memory."""
        if self._pool is None:
            for t in texts:
                yield self.g2p(t)
            return

        def chunks():
            buf: List[str] = []
            for t in texts:
                buf.append(t)
                if len(buf) == self.chunk:
                    yield buf
                    buf = []
            if buf:
                yield buf

        for part in self._pool.imap(_work, chunks()):
            yield from part

    def throughput(self, texts: Sequence[str]) -> float:
        """lines / s of ``map`` on this host (for sizing the pool against the GPU's utterance rate)."""
        t0 = time.perf_counter()
        self.map(texts)
        return len(texts) / max(time.perf_counter() - t0, 1e-9)

    def close(self):
        if self._pool is not None:
            self._pool.close()
            self._pool.join()
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
