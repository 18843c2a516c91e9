#!/usr/bin/env python3
"""Micro-benchmark of the conv-GEMM kernel on the layer shapes of BASELINE config 2 (B=32 x 1024 frames).
    python tools/bench_gemm.py [--shapes a,b,...] [--iters 10]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi  # noqa: E402

RF = 33024   # frame rows of config 2 incl. gaps (multiple of 256)
SHAPES = {
    # name: (dtype, M, K, N, taps, dil, residual, prologue)
    "s0_k3": (0, RF * 8, 256, 256, 3, 1, False, True),
    "s0_k7": (0, RF * 8, 256, 256, 7, 3, False, True),
    "s0_k11": (0, RF * 8, 256, 256, 11, 5, False, True),
    "s0_k11_res": (0, RF * 8, 256, 256, 11, 1, True, False),
    "s0q_k11": (0, 262144, 256, 256, 11, 5, False, True),      # exactly 4 x 512 tiles of 256 x 128 (wave-quantisation probe)
    "s0q_k7": (0, 262144, 256, 256, 7, 3, False, True),
    "s1_k3": (0, RF * 64, 128, 128, 3, 1, False, True),
    "s1_k7": (0, RF * 64, 128, 128, 7, 3, False, True),
    "s1_k11": (0, RF * 64, 128, 128, 11, 5, False, True),
    "s1_k11_res": (0, RF * 64, 128, 128, 11, 1, True, False),
    "s2_k3": (0, RF * 128, 64, 64, 3, 1, False, True),
    "s2_k11": (0, RF * 128, 64, 64, 11, 5, False, True),
    "s2_k11_res": (0, RF * 128, 64, 64, 11, 1, True, False),
    "s1_k11_mrf": (0, RF * 64, 128, 128, 11, 1, "mrf", False),     # last conv of a stage: fp16 residual + two fp16 MRF addends + post leaky-relu
    "s2_k11_mrf": (0, RF * 128, 64, 64, 11, 1, "mrf", False),
    "s2_k7_res": (0, RF * 128, 64, 64, 7, 1, True, False),
    "s3_k3": (0, RF * 256, 32, 32, 3, 1, False, True),
    "s3_k11": (0, RF * 256, 32, 32, 11, 5, False, True),
    "s3_k11_res": (0, RF * 256, 32, 32, 11, 1, True, False),
    "up0": (0, RF, 512, 2048, 3, 1, False, False),
    "up1": (0, RF * 8, 256, 1024, 3, 1, False, False),
    "dec_qkv": (0, RF, 384, 1152, 1, 1, False, False),
    "dec_out": (0, RF, 384, 384, 1, 1, True, False),
    "dec_ffn1": (0, RF, 384, 1536, 3, 1, False, False),
    "dec_ffn1_gelu": (0, RF, 384, 1536, 3, 1, False, "gelu"),
    "dec_ffn2": (0, RF, 1536, 384, 3, 1, True, False),
    # split precision (dtype 2: fp32 activations, fp16 hi/lo weights, 3 MFMAs per product)
    "x3_s0_k3": (2, RF * 8, 256, 256, 3, 1, False, True),
    "x3_s0_k11": (2, RF * 8, 256, 256, 11, 5, False, True),
    "x3_s1_k3": (2, RF * 64, 128, 128, 3, 1, False, True),
    "x3_s1_k7": (2, RF * 64, 128, 128, 7, 3, False, True),
    "x3_s1_k11": (2, RF * 64, 128, 128, 11, 5, False, True),
    "x3_s1_k11_res": (2, RF * 64, 128, 128, 11, 1, True, False),
    "x3_s2_k3": (2, RF * 128, 64, 64, 3, 1, False, True),
    "x3_s2_k11": (2, RF * 128, 64, 64, 11, 5, False, True),
    "x3_s3_k3": (2, RF * 256, 32, 32, 3, 1, False, True),
    "x3_s3_k11": (2, RF * 256, 32, 32, 11, 5, False, True),
    "x3_up0": (2, RF, 512, 2048, 3, 1, False, True),
    "x3_up1": (2, RF * 8, 256, 1024, 3, 1, False, True),
    "x3_dec_qkv": (2, RF, 384, 1152, 1, 1, False, False),
    "x3_dec_ffn1_gelu": (2, RF, 384, 1536, 3, 1, False, "gelu"),
    "x3_dec_ffn2": (2, RF, 1536, 384, 3, 1, True, False),
    "x3_enc_ffn1": (2, 8448, 384, 1536, 3, 1, False, "gelu"),
    "enc_ffn1_f32": (1, 8448, 384, 1536, 3, 1, False, False),
    "enc_ffn2_f32": (1, 8448, 1536, 384, 3, 1, True, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dbg", default="0", help="comma list of ConvGemmParams::reserved0 tuning switches to compare (bit 0: s_setprio around the MFMA cluster of the 4-wave kernel, bit 2 (4): 4-wave kernel instead of the phased 8-wave one)")
    args = ap.parse_args()
    lib = _ffi.lib()
    for name in args.shapes.split(","):
        dtype, M, K, N, taps, dil, res, pro = SHAPES[name]
        tdt = torch.float16 if dtype == 0 else torch.float32
        es = 2 if dtype == 0 else 4
        a = torch.randn(M + 128, K, device="cuda", dtype=torch.float32).to(tdt)
        w = (torch.randn(N, taps, K, device="cuda") / (K * taps) ** 0.5)
        w_lo = ((w - w.half().float()) * 2048.0).half() if dtype == 2 else None
        w = w.half() if dtype == 2 else w.to(tdt)
        bias = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda").to(torch.float16 if dtype == 0 else torch.float32) if res else None
        out = torch.empty(M, N, device="cuda", dtype=torch.float16 if dtype == 0 else torch.float32)
        d = _ffi.ev_conv_gemm_desc()
        d.dtype = dtype
        d.A, d.lda, d.W, d.bias = a[64:].data_ptr(), K, w.data_ptr(), bias.data_ptr()
        if w_lo is not None:
            d.W_lo = w_lo.data_ptr()
        d.M, d.N, d.K, d.taps, d.dil, d.center = M, N, K, taps, dil, (taps - 1) // 2
        d.out_scale = 1.0
        if pro == "gelu":
            d.act = 2
        elif pro:
            d.pro_lrelu, d.pro_slope, d.act, d.act_slope = 1, 0.1, 3, 0.1
        if r is not None:
            d.res, d.res_dtype, d.ldres = r.data_ptr(), (0 if dtype == 0 else 1), N
        if res == "mrf":
            ma, mb = torch.randn(M, N, device="cuda").half(), torch.randn(M, N, device="cuda").half()
            d.add16_a, d.add16_b, d.ldadd = ma.data_ptr(), mb.data_ptr(), N
            d.out_scale, d.post_lrelu, d.post_slope = 1.0 / 3.0, 1, 0.1
        if dtype == 0:
            d.out16 = out.data_ptr()
        else:
            d.out32 = out.data_ptr()
        d.ldo = N
        first = None
        for dbg in [int(x) for x in args.dbg.split(",")]:
            d.reserved0 = dbg
            out.zero_()
            for _ in range(2):
                lib.ev_op_conv_gemm(C.byref(d), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                lib.ev_op_conv_gemm(C.byref(d), None)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            flops = 2.0 * M * N * K * taps
            byt = M * K * es + M * N * es * (2 if res else 1)
            # variants selected by the switches must agree bit for bit (same accumulation order, same epilogue code)
            if first is None:
                first, same = out.clone(), ""
            else:
                nbad = int((out.view(torch.int16 if dtype == 0 else torch.int32) != first.view(torch.int16 if dtype == 0 else torch.int32)).sum())
                same = "  bits == first" if nbad == 0 else "  %d ELEMENTS DIFFER from first (max |d| %.3g)" % (nbad, float((out.float() - first.float()).abs().max()))
            print("%-14s dbg=%2d M=%8d K=%4d N=%4d taps=%2d dil=%d  %8.1f us  %7.1f TF/s  %6.2f TB/s%s" %
                  (name, dbg, M, K, N, taps, dil, ms * 1e3, flops / ms / 1e9, byt / ms / 1e9, same), flush=True)
        del a, w, out, r


if __name__ == "__main__":
    main()
