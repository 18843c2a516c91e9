#!/usr/bin/env python3
"""In-process A/B of engine switches on BASELINE configs[1] (32 x 256 phonemes): two handles on one device that differ in ONE EVEngine keyword, timed in
alternating rounds so that box-to-box and thermal drift (+-4 % between gpurun boxes) cancels.

    python tools/bench_ab.py --key mx_group --a True --b False [--rounds 4] [--steps 6]
"""
import argparse
import ast
import os
import sys
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
    ap.add_argument("--key", required=True)
    ap.add_argument("--a", required=True)
    ap.add_argument("--b", required=True)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--phonemes", type=int, default=256)
    a = ap.parse_args()
    args = types.SimpleNamespace(mode="am_vocoder", batch=a.batch, sub_batches=1, phonemes=a.phonemes)
    blob = broadcast_blob(0, 1, 0, None, dur_mode="bench")
    dev = torch.device("cuda", 0)
    works = []
    for v in (a.a, a.b):
        try:
            val = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            val = v
        e = EVEngine(device_id=0, **{a.key: val})
        e.load_blob_device(blob.data_ptr(), blob.numel(), keepalive=blob)
        works.append((val, bench.Workload(args, e, 0, dev, torch, _ffi)))
    outs = []
    for val, w in works:
        for _ in range(2):
            w.step()
        torch.cuda.synchronize()
    res = {0: [], 1: []}
    for r in range(a.rounds):
        for i, (val, w) in enumerate(works):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                w.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            res[i].append(ms)
            print("round %d  %s = %-8r %8.3f ms / step" % (r, a.key, val, ms), flush=True)
    for i, (val, w) in enumerate(works):
        print("%s = %-8r median %.3f ms / step (min %.3f)" % (a.key, val, float(np.median(res[i])), min(res[i])))
    print("B / A = %.4f" % (float(np.median(res[1])) / float(np.median(res[0]))))


if __name__ == "__main__":
    main()
