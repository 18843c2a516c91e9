#!/usr/bin/env python3
"""Times the fused ResBlock-pair kernels on the shapes of the B = 32 x 256-phoneme forward (tuning tool).

    python tools/bench_pair.py [--cases c32_k3,c32_k7,c32_k11,c64_k3]"""
import argparse
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi  # noqa: E402

RF = 33024
CASES = {"c32_k3": (32, RF * 256, 3, 3), "c32_k7": (32, RF * 256, 7, 3), "c32_k11": (32, RF * 256, 11, 5), "c64_k3": (64, RF * 128, 3, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="c32_k3,c32_k7,c32_k11,c64_k3")
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--dbg", default="0")
    a = ap.parse_args()
    lib = _ffi.lib()
    for name in a.cases.split(","):
        Cc, M, k, dil = CASES[name]
        x = torch.randn(M + 128, Cc, device="cuda").half()
        w1 = (torch.randn(Cc, k, Cc, device="cuda") / math.sqrt(Cc * k)).half()
        w2 = (torch.randn(Cc, k, Cc, device="cuda") / math.sqrt(Cc * k)).half()
        b1, b2 = torch.randn(Cc, device="cuda") * 0.1, torch.randn(Cc, device="cuda") * 0.1
        out = torch.empty(M, Cc, device="cuda", dtype=torch.float16)
        d = _ffi.ev_res_pair_desc()
        d.x, d.ldx, d.w1, d.b1, d.w2, d.M, d.k, d.dil = x[64:].data_ptr(), Cc, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), M, k, dil
        e = d.epi
        e.bias, e.res, e.res_dtype, e.ldres = b2.data_ptr(), x[64:].data_ptr(), 0, Cc
        e.out_scale, e.out16, e.ldo = 1.0, out.data_ptr(), Cc
        fn = lib.ev_op_resblock_pair_c32 if Cc == 32 else lib.ev_op_resblock_pair_c64
        first = None
        for dbg in [int(v) for v in a.dbg.split(",")]:
            e.reserved0 = dbg              # bit 2 (4): one block per CU (the k = 3 kernel's two-blocks-per-CU variant off)
            out.zero_()
            for _ in range(2):
                assert fn(C.byref(d), None) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn(C.byref(d), None)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            byt = 2.0 * M * Cc * 2
            if first is None:
                first, same = out.clone(), ""
            else:
                nbad = int((out.view(torch.int16) != first.view(torch.int16)).sum())
                same = "  bits == first" if nbad == 0 else "  %d ELEMENTS DIFFER" % nbad
            print("%-8s dbg=%d M=%9d C=%2d k=%2d dil=%d  %8.1f us  %6.2f TB/s (x in + out)  %7.1f TF/s%s" %
                  (name, dbg, M, Cc, k, dil, ms * 1e3, byt / ms / 1e9, 2 * 2.0 * M * Cc * Cc * k / ms / 1e9, same), flush=True)
        del x, out


if __name__ == "__main__":
    main()
