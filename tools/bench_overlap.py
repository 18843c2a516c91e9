#!/usr/bin/env python3
"""Do an MFMA-bound and an HBM-bound conv-GEMM overlap when they are launched on two streams?  (tuning probe)

    python tools/bench_overlap.py --a s1_k11 --b s1_k3

Prints the time of A alone, B alone, A then B on one stream, and A || B on two streams (per iteration, averaged).
Measured (round 2): no -- 937 vs 948 us for s1_k11 + s1_k3; and ONE launch with the tiles of both convs interleaved (so that every CU
holds one workgroup of each) is 13 % SLOWER than the two launches back to back (1 069 us): see DESIGN.md."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi  # noqa: E402
from bench_gemm import SHAPES  # noqa: E402


def make(name):
    dtype, M, K, N, taps, dil, res, pro = SHAPES[name]
    a = torch.randn(M + 128, K, device="cuda").half()
    w = (torch.randn(N, taps, K, device="cuda") / (K * taps) ** 0.5).half()
    bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").half() if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    d = _ffi.ev_conv_gemm_desc()
    d.dtype = 0
    d.A, d.lda, d.W, d.bias = a[64:].data_ptr(), K, w.data_ptr(), bias.data_ptr()
    d.M, d.N, d.K, d.taps, d.dil, d.center = M, N, K, taps, dil, (taps - 1) // 2
    d.out_scale = 1.0
    if pro:
        d.pro_lrelu, d.pro_slope, d.act, d.act_slope = 1, 0.1, 3, 0.1
    if r is not None:
        d.res, d.res_dtype, d.ldres = r.data_ptr(), 0, N
    d.out16, d.ldo = out.data_ptr(), N
    return d, (a, w, bias, r, out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="s1_k11")
    ap.add_argument("--b", default="s1_k3")
    ap.add_argument("--iters", type=int, default=6)
    args = ap.parse_args()
    lib = _ffi.lib()
    da, keep_a = make(args.a)
    db, keep_b = make(args.b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(which):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s1.wait_event(e0)
        s2.wait_event(e0)
        for _ in range(args.iters):
            if which in ("a", "ab_serial"):
                lib.ev_op_conv_gemm(C.byref(da), C.c_void_p(s1.cuda_stream))
            if which == "b":
                lib.ev_op_conv_gemm(C.byref(db), C.c_void_p(s1.cuda_stream))
            if which == "ab_serial":
                lib.ev_op_conv_gemm(C.byref(db), C.c_void_p(s1.cuda_stream))
            if which == "ab_parallel":
                lib.ev_op_conv_gemm(C.byref(da), C.c_void_p(s1.cuda_stream))
                lib.ev_op_conv_gemm(C.byref(db), C.c_void_p(s2.cuda_stream))
        d1, d2 = torch.cuda.Event(), torch.cuda.Event()
        d1.record(s1)
        d2.record(s2)
        torch.cuda.current_stream().wait_event(d1)
        torch.cuda.current_stream().wait_event(d2)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters * 1e3

    for which in ("a", "b", "ab_serial", "ab_parallel", "a", "b", "ab_serial", "ab_parallel"):
        run(which)                     # warm-up of this pattern
        print("%-12s %s / %s  %8.1f us per iteration" % (which, args.a, args.b, run(which)), flush=True)


if __name__ == "__main__":
    main()
