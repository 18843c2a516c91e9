#!/usr/bin/env python3
"""B = 1 latency breakdown (tuning probe): per-kernel-family time of one 64-phoneme utterance in each precision mode (profiled call: hipEvents
around every launch, single stream) beside the host-to-host wall time of the unprofiled call.    python tools/probe_b1.py [--phonemes 64]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd.engine import EVEngine  # noqa: E402
from emotivoice_amd.packer import pack_state_dict  # noqa: E402
from emotivoice_amd.synthetic import synth_inputs, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phonemes", type=int, default=64)
    args = ap.parse_args()
    blob, man = pack_state_dict(synth_state_dict(0, "bench"))
    u = synth_inputs(99, [args.phonemes], None)[0]
    for prec in ("mx", "fast", "strict"):
        eng = EVEngine(precision=prec)
        eng.load_blob(blob, man)
        best = 1e9
        for it in range(15):
            t0 = time.perf_counter()
            eng.synthesize([u])
            if it >= 3:
                best = min(best, time.perf_counter() - t0)
        eng.set_profiling(True)
        eng.synthesize([u])
        st = sorted(eng.kernel_stats(), key=lambda s: -s["ms"])
        eng.set_profiling(False)
        print("%-6s wall %.3f ms (numpy API incl. D2H); kernels %.3f ms in %d launches" % (prec, best * 1e3, sum(s["ms"] for s in st), sum(s["launches"] for s in st)))
        for s in st[:8]:
            print("       %-28s %3d launches %7.3f ms" % (s["name"], s["launches"], s["ms"]))
        eng.close()


if __name__ == "__main__":
    main()
