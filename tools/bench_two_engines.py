#!/usr/bin/env python3
"""Upper bound of what pipelining consecutive batches could buy (tuning probe): two engine handles on ONE device, each synthesising its
own B = 32 x 256-phoneme batch from its own host thread (ctypes releases the GIL; every handle has its own streams), against one handle
doing the same number of batches back to back.  If the low-occupancy token-rate kernels of one batch filled the gaps of the other's
vocoder, the pair would finish in less than twice the single time."""
import argparse
import os
import sys
import threading
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from emotivoice_amd import _ffi  # noqa: E402
from emotivoice_amd.engine import EVEngine  # noqa: E402
from emotivoice_amd.sharding import broadcast_blob  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--precision", default="mx")
    ap.add_argument("--split", action="store_true", help="ONE batch of 32 as two half-batches of 16 on two handles (intra-step pipelining) against one handle with 32")
    a = ap.parse_args()
    args = types.SimpleNamespace(mode="am_vocoder", batch=16 if a.split else 32, sub_batches=1, phonemes=256)
    args32 = types.SimpleNamespace(mode="am_vocoder", batch=32, sub_batches=1, phonemes=256)
    blob = broadcast_blob(0, 1, 0, None, dur_mode="bench")
    dev = torch.device("cuda", 0)
    engs, works = [], []
    for i in range(2):
        e = EVEngine(device_id=0, precision=a.precision)
        e.load_blob_device(blob.data_ptr(), blob.numel(), keepalive=blob)
        engs.append(e)
        works.append(bench.Workload(args, e, i, dev, torch, _ffi))

    def run(w, n, out):
        f = 0
        for _ in range(n):
            for c in w.calls:
                f += int(c().total_frames)
        out.append(f)

    for w in works:
        run(w, 2, [])
    torch.cuda.synchronize()
    cases = (("one handle, 2n batches", [(works[0], 2 * a.steps)]), ("two handles, n batches each", [(works[0], a.steps), (works[1], a.steps)]),
             ("one handle, 2n batches", [(works[0], 2 * a.steps)]), ("two handles, n batches each", [(works[0], a.steps), (works[1], a.steps)]))
    if a.split:
        e32 = EVEngine(device_id=0, precision=a.precision)
        e32.load_blob_device(blob.data_ptr(), blob.numel(), keepalive=blob)
        w32 = bench.Workload(args32, e32, 0, dev, torch, _ffi)
        run(w32, 2, [])
        cases = (("one handle, n batches of 32", [(w32, a.steps)]), ("two handles, n half-batches of 16 each", [(works[0], a.steps), (works[1], a.steps)]),
                 ("one handle, n batches of 32", [(w32, a.steps)]), ("two handles, n half-batches of 16 each", [(works[0], a.steps), (works[1], a.steps)]))
    for label, pairs in cases:
        outs = []
        ths = [threading.Thread(target=run, args=(w, n, outs)) for w, n in pairs]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-30s %8.1f ms total  %10.0f mel-frames/s" % (label, dt * 1e3, sum(outs) / dt), flush=True)


if __name__ == "__main__":
    main()
