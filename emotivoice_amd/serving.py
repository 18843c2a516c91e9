"""Serving shell on top of the FFI: dynamic batching + an OpenAI-compatible ``/v1/audio/speech`` endpoint.

Reference: openaiapi.py:111-184 -- one synchronous generator call per HTTP request (B = 1), speed changed afterwards by
pyrubberband time-stretching (:170-171).  Here:
  * requests from concurrent clients are collected by ``DynamicBatcher`` for at most ``max_wait_ms`` (or until ``max_batch`` /
    ``max_tokens``) and synthesised in ONE ev_synthesize call -- the engine evaluates every utterance with B = 1 semantics, so
    batching changes latency and throughput, never the audio (tests/test_gpu_parity.py::test_batch_invariance_bit_exact);
  * ``speed`` maps to the model's own duration scale alpha = 1 / speed (GaussianUpsampling's alpha, modules/alignment.py:183)
    instead of a time-stretch of the finished waveform; requests with different speeds form separate batches;
  * the text front-end (G2P) and the tokenizer are host-side callables supplied by the caller (the reference's frontend.py /
    AutoTokenizer), the style encoder is the device SimBERT (emotivoice_amd/simbert.py) or any ``text -> 768-vector`` callable.
Response formats: ``wav`` (16-bit PCM in a RIFF container) and ``pcm`` (raw little-endian int16); mp3 needs pydub / ffmpeg,
which are not part of this package.
"""
from __future__ import annotations

import io
import logging
import queue
import threading
import time
import wave
from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .text_io import wav_float_to_int16


_log = logging.getLogger("emotivoice_amd.serving")


@dataclass
class SynthesisRequest:
    ling: np.ndarray                 # (N,) int64 phoneme ids
    speaker: int
    style: object                    # (768,) array, or the prompt TEXT when the batcher owns the embedder (embed_batch_fn)
    content: object                  # (768,) array, or the content text
    alpha: float = 1.0               # duration scale (1 / speed)
    future: Future = field(default_factory=Future)
    t_submit: float = field(default_factory=time.perf_counter)


class DynamicBatcher:
    """Collects requests into batches for ``synth_fn(utts, alpha) -> list of float waveforms`` (one per utterance, in order).
    One worker thread owns the engine handle (the handle is not thread-safe, include/evhip.h): everything that touches the
    handle runs on that thread -- including the style / content embedding when the embedder lives on the same handle
    (``embed_batch_fn(texts) -> (len(texts), 768)``: requests then carry the TEXTS and the worker embeds a whole batch at once,
    which also batches the BERT forward)."""

    def __init__(self, synth_fn: Callable[[List[dict], float], Sequence[np.ndarray]], max_batch: int = 32, max_wait_ms: float = 5.0,
                 max_tokens: int = 16384, embed_batch_fn: Optional[Callable[[List[str]], np.ndarray]] = None,
                 n_vocab: Optional[int] = None, n_speaker: Optional[int] = None, max_len: int = 4096):
        self.synth_fn, self.max_batch, self.max_wait, self.max_tokens = synth_fn, max_batch, max_wait_ms * 1e-3, max_tokens
        self.embed_batch_fn, self.n_vocab, self.n_speaker, self.max_len = embed_batch_fn, n_vocab, n_speaker, max_len
        self._q: "queue.Queue[Optional[SynthesisRequest]]" = queue.Queue()
        self._carry: List[SynthesisRequest] = []
        self.batches: List[int] = []          # sizes of the batches formed so far (observability / tests)
        self.retried: int = 0                 # batches that failed as a whole and were re-run one request at a time
        self.loop_errors: int = 0             # exceptions the worker loop itself swallowed to stay alive (should stay 0)
        self._stop = False
        self._lock = threading.Lock()         # orders submit's (closed? -> enqueue) against close's (closed := True -> sentinel)
        self._thread = threading.Thread(target=self._loop, name="ev-batcher", daemon=True)
        self._thread.start()

    def submit(self, ling, speaker: int, style, content, alpha: float = 1.0) -> Future:
        """Validation happens HERE (ValueError -> the handler's 400), so that one malformed request cannot take the requests
        batched with it down: the reference serves every request on its own (openaiapi.py:159-184)."""
        ling = np.asarray(ling, np.int64).reshape(-1)
        if ling.size == 0:
            raise ValueError("empty phoneme sequence")
        if ling.size > self.max_len:
            raise ValueError("phoneme sequence too long (%d > %d)" % (ling.size, self.max_len))
        if ling.min() < 0 or (self.n_vocab is not None and ling.max() >= self.n_vocab):
            raise ValueError("phoneme id out of range")
        if int(speaker) < 0 or (self.n_speaker is not None and int(speaker) >= self.n_speaker):
            raise ValueError("speaker id out of range")
        if not (alpha > 0 and np.isfinite(alpha)):
            raise ValueError("alpha must be positive")

        def emb(x):
            if isinstance(x, str):
                if self.embed_batch_fn is None:
                    raise ValueError("text embeddings need a batcher with embed_batch_fn")
                return x
            x = np.asarray(x, np.float32).reshape(-1)
            if not np.isfinite(x).all():
                raise ValueError("embedding is not finite")
            return x
        req = SynthesisRequest(ling, int(speaker), emb(style), emb(content), float(alpha))
        with self._lock:
            if self._stop:
                raise RuntimeError("batcher is closed")
            self._q.put(req)
        return req.future

    def close(self, timeout: float = 30.0) -> bool:
        """Stop accepting requests, let the worker drain what it has, and fail whatever is left.  Returns True when the worker has exited.
        False = the join timed out: the worker is still inside ``synth_fn`` (a hung device call, say).  Its queued and carried requests then stay
        with it -- it fails or serves them if it ever returns; a client must not wait on such a Future without its own timeout (``TTSService.speech``
        passes one), and the owner of the batcher decides what a wedged engine means for the process (the reference's server has no such state:
        one synchronous call per request, openaiapi.py:159-184).  ``close`` can be called again later to collect the worker."""
        with self._lock:
            if not self._stop:
                self._stop = True
                self._q.put(None)
        self._thread.join(timeout=timeout)
        if self._thread.is_alive():
            # the worker is still inside a batch (join timed out): it owns the queue and _carry; the sentinel stays queued, so it exits after
            # draining what it has, and nothing is taken from under it here
            _log.error("DynamicBatcher.close: the worker did not finish within %.1f s; %d request(s) stay queued behind it", timeout, self._q.qsize())
            return False
        # the worker has exited (or died): whatever is still queued or carried must not leave its client waiting
        left = list(self._carry)
        self._carry = []
        while True:
            try:
                r = self._q.get_nowait()
            except queue.Empty:
                break
            if r is not None:
                left.append(r)
        for r in left:
            _resolve(r.future, exception=RuntimeError("batcher closed"))
        return True

    def _take_batch(self) -> List[SynthesisRequest]:
        """Block for the first request, then keep collecting until the batch is full, the token budget is reached or
        ``max_wait`` has passed since the first request arrived.  Requests with another alpha are carried to the next batch."""
        first = self._carry.pop(0) if self._carry else self._q.get()
        if first is None:
            return []
        batch = [first]
        cur = first                                         # the request whose fields are being looked at (the one to blame if that raises)
        try:
            tokens = len(first.ling)
            deadline = time.perf_counter() + self.max_wait
            keep = []
            for r in self._carry:                           # carried requests first (they have waited longest)
                cur = r
                if r.alpha == first.alpha and len(batch) < self.max_batch and tokens + len(r.ling) <= self.max_tokens:
                    batch.append(r); tokens += len(r.ling)
                else:
                    keep.append(r)
            self._carry = keep
            while len(batch) < self.max_batch and tokens < self.max_tokens:
                timeout = deadline - time.perf_counter()
                if timeout <= 0:
                    break
                try:
                    r = self._q.get(timeout=timeout)
                except queue.Empty:
                    break
                if r is None:
                    self._q.put(None)
                    break
                batch.append(r)                             # (taken off the queue: from here on it is somebody's responsibility)
                cur = r
                if r.alpha == first.alpha and tokens + len(r.ling) <= self.max_tokens:
                    tokens += len(r.ling)
                else:
                    batch.pop()
                    self._carry.append(r)
            return batch
        except Exception:
            # a malformed request (e.g. no usable ``ling``) blew up the collection: what was already taken off the queue / the carry list goes
            # back to the carry list, minus the offender, which fails alone -- nobody is left waiting
            rest = [r for r in batch if r is not cur]
            self._carry = rest + [r for r in self._carry if r is not cur and all(r is not x for x in rest)]
            _resolve(cur.future, exception=RuntimeError("request could not be batched (malformed fields)"))
            raise

    def _run(self, batch: List[SynthesisRequest]):
        if self.embed_batch_fn is not None:
            texts = sorted({x for r in batch for x in (r.style, r.content) if isinstance(x, str)})
            if texts:
                vec = np.asarray(self.embed_batch_fn(texts), np.float32)
                if vec.shape[0] != len(texts):
                    raise RuntimeError("embed_batch_fn returned %d vectors for %d texts" % (vec.shape[0], len(texts)))
                table = {t: vec[i] for i, t in enumerate(texts)}
                for r in batch:
                    r.style = table[r.style] if isinstance(r.style, str) else r.style
                    r.content = table[r.content] if isinstance(r.content, str) else r.content
        wavs = self.synth_fn([dict(ling=r.ling, speaker=r.speaker, style=r.style, content=r.content) for r in batch], batch[0].alpha)
        if len(wavs) != len(batch):
            raise RuntimeError("synth_fn returned %d waveforms for %d requests" % (len(wavs), len(batch)))
        for r, w in zip(batch, wavs):
            _resolve(r.future, result=np.array(w, np.float32, copy=True))

    @staticmethod
    def _claim(r: "SynthesisRequest") -> bool:
        """True if the request's Future could be moved to RUNNING.  A Future its client has cancelled, or already resolved behind the
        batcher's back (set_exception / set_result as the client's own timeout handling -- set_running_or_notify_cancel raises on a
        FINISHED Future), drops ONLY itself: the requests co-batched with it must still be served."""
        try:
            return (not r.future.done()) and r.future.set_running_or_notify_cancel()
        except Exception:
            return False

    def _loop(self):
        consecutive = 0
        while True:
            batch: List[SynthesisRequest] = []
            try:
                batch = self._take_batch()
                if not batch:
                    if self._stop and not self._carry:
                        return
                    continue
                # a client may cancel or resolve its Future while it waits: such requests are dropped here one by one, and a Future that is
                # cancelled later (mid-batch) is simply not resolved (_resolve) -- neither may take the worker thread or its batch-mates down
                batch = [r for r in batch if self._claim(r)]
                if not batch:
                    continue
                self.batches.append(len(batch))
                try:
                    self._run(batch)
                except Exception as e:
                    if len(batch) == 1:
                        _resolve(batch[0].future, exception=e)
                        continue
                    # a batch failed as a whole: re-run its requests one at a time so that only the offender fails
                    self.retried += 1
                    for r in batch:
                        if r.future.done():
                            continue
                        try:
                            self._run([r])
                        except Exception as e1:
                            _resolve(r.future, exception=e1)
                consecutive = 0
            except Exception as e:     # nothing a single request does may end the thread every later request depends on
                self.loop_errors += 1
                consecutive += 1
                _log.exception("DynamicBatcher worker: unexpected error (%d in a row)", consecutive)
                for r in batch:        # ... and nobody may be left waiting on a batch that was dropped half-way
                    _resolve(r.future, exception=e)
                time.sleep(min(0.5, 0.01 * consecutive))      # a persistent failure (e.g. in _take_batch) must not spin a core


def _resolve(future: Future, result=None, exception: Optional[BaseException] = None):
    """set_result / set_exception that tolerates a Future its client has cancelled or that was resolved already."""
    try:
        if future.done():
            return
        if exception is not None:
            future.set_exception(exception)
        else:
            future.set_result(result)
    except Exception:            # concurrent.futures.InvalidStateError (cancelled between the check and the set)
        pass


def engine_synth_fn(engine) -> Callable[[List[dict], float], Sequence[np.ndarray]]:
    """``synth_fn`` of a DynamicBatcher for an EVEngine."""
    return lambda utts, alpha: engine.synthesize(utts, alpha=alpha)["wav_list"]


def engine_embed_batch_fn(engine, tokenize: Callable[[str], Sequence[int]]) -> Callable[[List[str]], np.ndarray]:
    """``embed_batch_fn`` of a DynamicBatcher for the device SimBERT living on ``engine`` (ev_style_embed on the SAME handle as
    ev_synthesize -- which is why it has to run on the batcher's thread): ``tokenize(text) -> [CLS] ... [SEP]`` ids
    (emotivoice_amd.wordpiece or the reference's AutoTokenizer, simbert.py / inference_am_vocoder_joint.py:25-38)."""
    return lambda texts: engine.style_embed([np.asarray(tokenize(t), np.int64) for t in texts])


def encode_audio(wav_f32: np.ndarray, response_format: str, sample_rate: int) -> bytes:
    pcm = wav_float_to_int16(wav_f32)              # the reference's int16 epilogue (openaiapi.py:147-148)
    if response_format == "pcm":
        return pcm.tobytes()
    if response_format == "wav":
        buf = io.BytesIO()
        with wave.open(buf, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sample_rate)
            w.writeframes(pcm.tobytes())
        return buf.getvalue()
    raise ValueError("response_format %r is not available (wav, pcm); mp3 needs pydub / ffmpeg" % response_format)


class TTSService:
    """text -> audio: G2P, token / speaker lookup, style + content embedding, batched synthesis.  All host-side pieces are
    injected: ``g2p(text) -> phoneme string`` (the reference's frontend.g2p_cn_en), ``embed(text) -> (768,)`` (the device SimBERT
    through a tokenizer, or a placeholder)."""

    def __init__(self, batcher: DynamicBatcher, token2id: Dict[str, int], speaker2id: Dict[str, int], g2p: Callable[[str], str],
                 embed: Optional[Callable[[str], np.ndarray]] = None, sample_rate: int = 16000):
        """``embed`` runs on the CALLER's thread (FastAPI's handler pool): it must be thread-safe and must not share an engine
        handle with the batcher.  For the device SimBERT on the generator's own handle leave it None and give the BATCHER an
        ``embed_batch_fn``: the texts then travel with the request and are embedded on the worker thread, a batch at a time."""
        self.batcher, self.token2id, self.speaker2id, self.g2p, self.embed, self.sample_rate = batcher, token2id, speaker2id, g2p, embed, sample_rate
        if embed is None and batcher.embed_batch_fn is None:
            raise ValueError("TTSService needs embed= or a batcher with embed_batch_fn=")

    def submit(self, text: str, voice: str, prompt: str = "", speed: float = 1.0) -> Future:
        if not (0.25 <= speed <= 4.0):
            raise ValueError("speed must be within [0.25, 4]")
        ling = np.array([self.token2id[ph] for ph in self.g2p(text).split()], np.int64)       # KeyError like openaiapi.py:128
        if ling.size == 0:
            raise ValueError("input has no phonemes")
        if self.embed is None:
            return self.batcher.submit(ling, self.speaker2id[voice], prompt, text, alpha=1.0 / speed)
        return self.batcher.submit(ling, self.speaker2id[voice], self.embed(prompt), self.embed(text), alpha=1.0 / speed)

    def speech(self, text: str, voice: str, prompt: str = "", speed: float = 1.0, response_format: str = "wav", timeout: float = 120.0) -> bytes:
        return encode_audio(self.submit(text, voice, prompt, speed).result(timeout=timeout), response_format, self.sample_rate)


try:          # the request schema lives at module level: FastAPI resolves the handler's annotations in the module namespace
    from pydantic import BaseModel as _BaseModel

    class SpeechRequest(_BaseModel):
        """openaiapi.py:150-157 (default response_format here is wav: mp3 needs pydub / ffmpeg)."""
        input: str
        voice: str = "8051"
        prompt: Optional[str] = ""
        language: Optional[str] = "zh_us"
        model: Optional[str] = "emoti-voice"
        response_format: Optional[str] = "wav"
        speed: Optional[float] = 1.0
except ImportError:          # pydantic / fastapi are only needed by create_app
    SpeechRequest = None


def create_app(service: TTSService):
    """FastAPI application with the reference's route and request schema (openaiapi.py:150-184)."""
    from fastapi import FastAPI, HTTPException, Response

    app = FastAPI()

    @app.post("/v1/audio/speech")
    def text_to_speech(req: SpeechRequest):
        try:
            data = service.speech(req.input, req.voice, req.prompt or "", req.speed or 1.0, req.response_format or "wav")
        except KeyError as e:
            raise HTTPException(status_code=400, detail="unknown voice or phoneme: %s" % e)
        except ValueError as e:
            raise HTTPException(status_code=400, detail=str(e))
        return Response(content=data, media_type="audio/%s" % (req.response_format or "wav"))

    return app
