#!/usr/bin/env python3
"""Does running a ResBlock pair (conv1 -> tmp -> conv2 + residual) on row chunks small enough for the 256 MB Infinity Cache
beat running each layer over all rows?  (tuning probe: timing only, chunk-border halos are not made consistent)

    python tools/bench_chunked.py [--C 64 --k 7 --rows_per_frame 128]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi  # noqa: E402

RF = 33024


def desc(x, w, bias, out, M, Cc, taps, dil, res=None, pro=False):
    d = _ffi.ev_conv_gemm_desc()
    d.dtype = 0
    d.A, d.lda, d.W, d.bias = x, Cc, w.data_ptr(), bias.data_ptr()
    d.M, d.N, d.K, d.taps, d.dil, d.center = M, Cc, Cc, taps, dil, (taps - 1) // 2
    d.out_scale = 1.0
    if pro:
        d.pro_lrelu, d.pro_slope, d.act, d.act_slope = 1, 0.1, 3, 0.1
    if res is not None:
        d.res, d.res_dtype, d.ldres = res, 0, Cc
    d.out16, d.ldo = out, Cc
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--k", type=int, default=7)
    ap.add_argument("--rows_per_frame", type=int, default=128)
    ap.add_argument("--pairs", type=int, default=3, help="consecutive pairs of one ResBlock run per chunk")
    args = ap.parse_args()
    lib = _ffi.lib()
    Cc, k = args.C, args.k
    M = RF * args.rows_per_frame
    x = torch.randn(M + 128, Cc, device="cuda").half()
    tmp = torch.empty(M + 128, Cc, device="cuda", dtype=torch.float16)
    ya = torch.empty(M + 128, Cc, device="cuda", dtype=torch.float16)
    yb = torch.empty(M + 128, Cc, device="cuda", dtype=torch.float16)
    w1 = (torch.randn(Cc, k, Cc, device="cuda") / (Cc * k) ** 0.5).half()
    w2 = (torch.randn(Cc, k, Cc, device="cuda") / (Cc * k) ** 0.5).half()
    b = torch.randn(Cc, device="cuda")
    es = 2 * Cc

    def run(nchunks):
        rows = M // nchunks // 256 * 256
        for c in range(nchunks):
            r0 = c * rows
            m = rows if c + 1 < nchunks else M - r0
            src, dsts = x, [ya, yb, ya]
            for j in range(args.pairs):
                o = (64 + r0) * es
                d1 = desc(src.data_ptr() + o, w1, b, tmp.data_ptr() + o, m, Cc, k, [1, 3, 5][j % 3], pro=True)
                lib.ev_op_conv_gemm(C.byref(d1), None)
                d2 = desc(tmp.data_ptr() + o, w2, b, dsts[j].data_ptr() + o, m, Cc, k, 1, res=src.data_ptr() + o)
                lib.ev_op_conv_gemm(C.byref(d2), None)
                src = dsts[j]

    for nchunks in (1, 2, 4, 8, 16, 32, 64):
        run(nchunks)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run(nchunks)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print("C=%d k=%d rows=%d pairs=%d chunks=%3d (%.0f MB per tensor chunk): %8.1f us" %
              (Cc, k, M, args.pairs, nchunks, M / nchunks * es / 1e6, ms * 1e3), flush=True)


if __name__ == "__main__":
    main()
