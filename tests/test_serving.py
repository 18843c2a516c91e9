"""Host logic of the serving shell and the front-end pool (no GPU: the engine is replaced by a fake synth_fn)."""
import io
import threading
import time
import wave

import numpy as np
import pytest

from emotivoice_amd.frontend_pool import FrontendPool
from emotivoice_amd.serving import DynamicBatcher, TTSService, create_app, encode_audio


def _fake_synth(log, delay=0.02):
    def fn(utts, alpha):
        log.append((len(utts), alpha))
        time.sleep(delay)                                  # the "GPU" is busy: later requests pile up meanwhile
        return [np.full(int(256 * len(u["ling"]) * alpha), float(u["speaker"]) / 100.0, np.float32) for u in utts]
    return fn


def test_dynamic_batcher_batches_routes_and_groups_by_alpha():
    log = []
    b = DynamicBatcher(_fake_synth(log), max_batch=8, max_wait_ms=30)
    futs = {}

    def client(i):
        futs[i] = b.submit(np.arange(3 + i % 5), speaker=i, style=np.zeros(768), content=np.zeros(768), alpha=2.0 if i % 7 == 0 else 1.0)

    ths = [threading.Thread(target=client, args=(i,)) for i in range(40)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for i in range(40):
        w = futs[i].result(timeout=10)
        alpha = 2.0 if i % 7 == 0 else 1.0
        assert w.shape == (int(256 * (3 + i % 5) * alpha),) and np.allclose(w, i / 100.0)      # each client gets ITS audio
    b.close()
    assert sum(n for n, _ in log) == 40 and max(n for n, _ in log) <= 8
    assert len(log) < 40                                   # requests were actually batched
    assert all(a in (1.0, 2.0) for _, a in log) and {a for _, a in log} == {1.0, 2.0}
    assert b.batches == [n for n, _ in log]


def test_batcher_failure_is_confined_to_its_batch():
    calls = []

    def fn(utts, alpha):
        calls.append(len(utts))
        if any(u["speaker"] == 13 for u in utts):
            raise RuntimeError("boom")
        return [np.zeros(4, np.float32) for _ in utts]

    b = DynamicBatcher(fn, max_batch=4, max_wait_ms=1)
    bad = b.submit(np.arange(3), 13, np.zeros(768), np.zeros(768))
    with pytest.raises(RuntimeError):
        bad.result(timeout=5)
    ok = b.submit(np.arange(3), 1, np.zeros(768), np.zeros(768))
    assert ok.result(timeout=5).shape == (4,)
    b.close()
    with pytest.raises(RuntimeError):
        b.submit(np.arange(3), 1, np.zeros(768), np.zeros(768))


def test_one_bad_request_does_not_fail_its_batch_mates():
    """a request the engine rejects must not 500 the unrelated requests batched with it -- the batch is re-run one
    request at a time and only the offender fails (the reference serves each request on its own)."""
    calls = []

    def fn(utts, alpha):
        calls.append(len(utts))
        if any(u["speaker"] == 13 for u in utts):
            raise RuntimeError("boom")
        return [np.full(4, u["speaker"], np.float32) for u in utts]

    b = DynamicBatcher(fn, max_batch=8, max_wait_ms=200)
    futs = [b.submit(np.arange(3), spk, np.zeros(768), np.zeros(768)) for spk in (1, 13, 2, 3)]
    assert [f.result(timeout=5)[0] for i, f in enumerate(futs) if i != 1] == [1.0, 2.0, 3.0]
    with pytest.raises(RuntimeError):
        futs[1].result(timeout=5)
    assert calls[0] == 4 and calls[1:] == [1, 1, 1, 1] and b.retried == 1
    b.close()


def test_submit_validates_before_batching():
    b = DynamicBatcher(lambda u, a: [np.zeros(4, np.float32) for _ in u], max_batch=4, max_wait_ms=1, n_vocab=10, n_speaker=5, max_len=8)
    z = np.zeros(768)
    for ling, spk in (([], 0), ([1, 2, 10], 0), ([-1], 0), ([1, 2], 5), ([1, 2], -1), (list(range(9)), 0)):
        with pytest.raises(ValueError):
            b.submit(np.array(ling, np.int64), spk, z, z)
    with pytest.raises(ValueError):
        b.submit(np.arange(3), 0, np.full(768, np.nan), z)
    with pytest.raises(ValueError):
        b.submit(np.arange(3), 0, "a prompt text", z)               # texts need embed_batch_fn
    assert b.submit(np.arange(3), 0, z, z).result(timeout=5).shape == (4,) and b.batches == [1]
    b.close()


def test_texts_are_embedded_on_the_worker_thread_one_call_per_batch():
    """the device SimBERT shares the engine handle with ev_synthesize, so embedding must not run on the callers'
    threads.  With embed_batch_fn the requests carry the texts; the worker embeds the batch's distinct texts in ONE call."""
    import threading
    seen = []

    def embed_batch(texts):
        seen.append((threading.current_thread().name, list(texts)))
        return np.stack([np.full(768, len(t), np.float32) for t in texts])

    got = []

    def fn(utts, alpha):
        got.append([(float(u["style"][0]), float(u["content"][0])) for u in utts])
        return [np.zeros(4, np.float32) for _ in utts]

    b = DynamicBatcher(fn, max_batch=8, max_wait_ms=200, embed_batch_fn=embed_batch)
    svc = TTSService(b, {"a": 1}, {"v": 0}, g2p=lambda t: "a " * len(t))
    futs = [svc.submit(t, "v", prompt="happy") for t in ("xx", "yyy", "xx")]
    [f.result(timeout=5) for f in futs]
    b.close()
    assert len(seen) == 1 and seen[0][0] == "ev-batcher" and seen[0][1] == ["happy", "xx", "yyy"]
    assert got == [[(5.0, 2.0), (5.0, 3.0), (5.0, 2.0)]]
    with pytest.raises(ValueError):
        TTSService(DynamicBatcher(fn), {"a": 1}, {"v": 0}, g2p=str)        # neither embed nor embed_batch_fn


def test_close_resolves_every_pending_future():
    import threading
    gate = threading.Event()

    def slow(utts, alpha):
        gate.wait(5)
        return [np.zeros(4, np.float32) for _ in utts]

    b = DynamicBatcher(slow, max_batch=1, max_wait_ms=1)
    futs = [b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768), alpha=1.0 + i) for i in range(4)]
    closer = threading.Thread(target=b.close)
    closer.start()
    gate.set()
    closer.join(10)
    assert not closer.is_alive()
    for f in futs:                  # served or failed, never left hanging
        assert f.done()
    with pytest.raises(RuntimeError):
        b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768))


def test_cancelled_future_does_not_kill_the_worker():
    """a client cancelling its Future while its batch runs made set_result raise InvalidStateError, the error path raised
    again on the same Future and the batcher thread died -- every later request hung."""
    import threading
    started, gate = threading.Event(), threading.Event()

    def slow(utts, alpha):
        started.set()
        gate.wait(5)
        return [np.full(4, len(u["ling"]), np.float32) for u in utts]

    b = DynamicBatcher(slow, max_batch=1, max_wait_ms=1)
    f1 = b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768))
    assert started.wait(5)
    f2 = b.submit(np.arange(5), 0, np.zeros(768), np.zeros(768))
    assert f2.cancel()                      # still queued: dropped when its batch is formed
    f1.cancel()                             # running: cancel() is refused, but the Future API allows the call
    gate.set()
    f3 = b.submit(np.arange(7), 0, np.zeros(768), np.zeros(768))
    assert f3.result(timeout=5)[0] == 7     # the worker is alive and serving
    assert f1.result(timeout=5)[0] == 3
    assert b._thread.is_alive() and b.loop_errors == 0
    b.close()


def test_worker_survives_a_future_resolved_behind_its_back():
    """A Future that is already resolved (e.g. failed by the caller's timeout handling) when the batch finishes must be skipped, and an
    exception from synth_fn on such a request must not end the thread either."""
    def boom(utts, alpha):
        raise RuntimeError("synth failed")

    b = DynamicBatcher(boom, max_batch=1, max_wait_ms=1)
    f = b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768))
    with pytest.raises(RuntimeError, match="synth failed"):
        f.result(timeout=5)
    g = b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768))
    with pytest.raises(RuntimeError, match="synth failed"):
        g.result(timeout=5)
    assert b._thread.is_alive()
    b.close()


def test_future_resolved_while_queued_drops_only_itself():
    """a client that fails its own Future (set_exception as its timeout handling) while the request is still QUEUED made
    set_running_or_notify_cancel raise inside the batch filter; the loop's catch-all swallowed it and the whole batch -- the healthy
    request co-batched with it included -- was never resolved.  Now such a request drops only itself."""
    started, gate = threading.Event(), threading.Event()

    def slow(utts, alpha):
        started.set()
        gate.wait(5)
        return [np.full(4, len(u["ling"]), np.float32) for u in utts]

    b = DynamicBatcher(slow, max_batch=4, max_wait_ms=20)
    f0 = b.submit(np.arange(2), 0, np.zeros(768), np.zeros(768))
    assert started.wait(5)                   # the worker is busy with f0: the next two requests wait in the queue together
    f1 = b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768))
    f2 = b.submit(np.arange(5), 0, np.zeros(768), np.zeros(768))
    f1.set_exception(TimeoutError("client gave up"))       # resolved behind the batcher's back, while still queued
    gate.set()
    assert f0.result(timeout=5)[0] == 2
    assert f2.result(timeout=5)[0] == 5      # the healthy batch-mate is served
    with pytest.raises(TimeoutError):
        f1.result(timeout=1)
    assert b.loop_errors == 0 and b._thread.is_alive()
    assert b.batches[-1] == 1                # the dropped request never reached synth_fn
    b.close()


def test_loop_error_fails_the_batch_it_dropped_and_backs_off(caplog):
    """Whatever the loop's catch-all swallows must (a) fail every unresolved Future of the batch in hand, (b) be logged, (c) not spin."""
    b = DynamicBatcher(lambda utts, alpha: [np.zeros(4, np.float32) for _ in utts], max_batch=2, max_wait_ms=1)
    orig = b._claim
    calls = []

    def broken_claim(r):
        calls.append(r)
        if len(calls) == 1:
            raise MemoryError("unexpected")   # stands for any error outside _run's own try block
        return orig(r)
    b._claim = broken_claim
    with caplog.at_level("ERROR", logger="emotivoice_amd.serving"):
        f = b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768))
        with pytest.raises(MemoryError):
            f.result(timeout=5)
    assert b.loop_errors == 1 and any("DynamicBatcher worker" in r.message for r in caplog.records)
    g = b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768))
    assert g.result(timeout=5).shape == (4,) and b._thread.is_alive()
    b.close()


def test_error_inside_take_batch_leaves_nobody_waiting():
    """an exception raised INSIDE _take_batch (after it has pulled requests off the queue into its local list) used to drop those requests
    unresolved -- the loop's catch-all only saw an empty batch.  Now the offender at the head fails alone and what was collected goes back to the carry list."""
    import threading
    from emotivoice_amd.serving import SynthesisRequest
    gate = threading.Event()

    def synth(utts, alpha):
        gate.wait(5)
        return [np.zeros(4, np.float32) for _ in utts]
    b = DynamicBatcher(synth, max_batch=4, max_wait_ms=200)
    first = b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768))          # occupies the worker until the gate opens

    class NoLen:                                     # a request whose ``ling`` cannot be measured: len() raises inside _take_batch
        def __len__(self):
            raise TypeError("no length")
    time.sleep(0.05)
    bad = SynthesisRequest.__new__(SynthesisRequest)
    bad.__dict__.update(ling=NoLen(), speaker=0, style=np.zeros(768), content=np.zeros(768), alpha=1.0)
    from concurrent.futures import Future
    bad.future = Future()
    b._q.put(bad)                                    # (submit() validates; this stands for any field that breaks the collection loop)
    good = [b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768)) for _ in range(2)]
    gate.set()
    assert first.result(timeout=5).shape == (4,)
    with pytest.raises(Exception):
        bad.future.result(timeout=5)
    for g in good:
        assert g.result(timeout=5).shape == (4,)     # collected behind the offender or still queued: served either way
    assert b._thread.is_alive()
    b.close()


def test_close_with_a_busy_worker_leaves_it_its_sentinel():
    """close() whose join timed out used to drain the queue -- sentinel included -- so the worker blocked forever after its
    batch and its carried requests were never failed.  Now a live worker keeps the queue: it finishes, sees the sentinel and exits."""
    import threading
    gate = threading.Event()

    def slow(utts, alpha):
        gate.wait(5)
        return [np.zeros(4, np.float32) for _ in utts]

    b = DynamicBatcher(slow, max_batch=1, max_wait_ms=1)
    futs = [b.submit(np.arange(3), 0, np.zeros(768), np.zeros(768), alpha=1.0 + i) for i in range(3)]
    assert b.close(timeout=0.05) is False   # the worker is inside the first batch: join times out, and close() says so
    assert b._thread.is_alive()
    gate.set()
    b._thread.join(5)
    assert not b._thread.is_alive()         # it drained its requests, met the sentinel and returned
    for f in futs:
        assert f.done()
    assert b.close() is True                # idempotent; the worker has been collected


def test_token_budget_splits_batches():
    log = []
    b = DynamicBatcher(_fake_synth(log, 0.0), max_batch=64, max_wait_ms=50, max_tokens=100)
    futs = [b.submit(np.arange(40), 0, np.zeros(768), np.zeros(768)) for _ in range(5)]
    [f.result(timeout=5) for f in futs]
    b.close()
    assert all(n <= 2 for n, _ in log) and sum(n for n, _ in log) == 5


def test_encode_audio_formats():
    x = np.array([0.0, 0.5, -0.5, 0.99997], np.float32)
    assert np.frombuffer(encode_audio(x, "pcm", 16000), np.int16).tolist() == [0, 16384, -16384, 32767]
    with wave.open(io.BytesIO(encode_audio(x, "wav", 16000))) as w:
        assert (w.getframerate(), w.getsampwidth(), w.getnchannels(), w.getnframes()) == (16000, 2, 1, 4)
    with pytest.raises(ValueError):
        encode_audio(x, "mp3", 16000)


def test_openai_compatible_endpoint():
    pytest.importorskip("fastapi")
    pytest.importorskip("httpx")
    from fastapi.testclient import TestClient
    log = []
    b = DynamicBatcher(_fake_synth(log, 0.0), max_batch=4, max_wait_ms=1)
    tok = {"<sos/eos>": 0, "a": 1, "b": 2}
    svc = TTSService(b, tok, {"8051": 0, "9000": 7}, g2p=lambda t: "<sos/eos> " + " ".join("a" if c < "n" else "b" for c in t if c.isalpha()) + " <sos/eos>",
                     embed=lambda t: np.zeros(768, np.float32))
    client = TestClient(create_app(svc))
    r = client.post("/v1/audio/speech", json={"input": "hello", "voice": "9000", "response_format": "wav", "speed": 2.0})
    assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
    with wave.open(io.BytesIO(r.content)) as w:
        assert w.getframerate() == 16000 and w.getnframes() == int(256 * 7 * 0.5)      # 7 phonemes, speed 2 -> alpha 0.5
    assert log[-1] == (1, 0.5)
    assert client.post("/v1/audio/speech", json={"input": "hello", "voice": "nobody"}).status_code == 400
    assert client.post("/v1/audio/speech", json={"input": "hello", "voice": "8051", "response_format": "mp3"}).status_code == 400
    assert client.post("/v1/audio/speech", json={"input": "hello", "voice": "8051", "speed": 9.0}).status_code == 400
    b.close()


def _toy_g2p(text):
    return "<sos/eos> " + " ".join("p%d" % (ord(c) % 50) for c in text if not c.isspace()) + " <sos/eos>"


def test_frontend_pool_keeps_order_and_matches_serial():
    texts = ["line %d with some words %s" % (i, "x" * (i % 13)) for i in range(500)]
    serial = [_toy_g2p(t) for t in texts]
    with FrontendPool(_toy_g2p, workers=4, chunk=16) as pool:
        assert pool.map(texts) == serial
        assert list(pool.stream(iter(texts))) == serial
        assert pool.throughput(texts) > 0
    with FrontendPool(_toy_g2p, workers=1) as pool:
        assert pool.map(texts[:20]) == serial[:20]
