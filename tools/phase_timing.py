#!/usr/bin/env python3
"""Where the waves of conv_gemm_phased_kernel spend their cycles (tuning tool; needs the timing build):

    python emotivoice_amd/csrc/build.py --variant timing EV_PH_TIMING
    EVHIP_LIB=emotivoice_amd/csrc/libevhip_timing.so python tools/phase_timing.py --shapes s1_k11,s1_k3

Every wave sums s_memtime deltas per part of a step in SGPRs and writes seven totals at the end (through the otherwise unused row_seq
pointer).  Reported per wave, averaged over all waves of the launch, split by phase group (waves 0-3 / 4-7)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi  # noqa: E402
from bench_gemm import SHAPES  # noqa: E402

NAMES = ["prologue", "vmcnt wait", "load phase", "barrier (load)", "matrix phase", "barrier (matrix)", "epilogue"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="s1_k11,s1_k7,s1_k3,s0_k11,dec_ffn2")
    args = ap.parse_args()
    lib = _ffi.lib()
    for name in args.shapes.split(","):
        dtype, M, K, N, taps, dil, res, pro = SHAPES[name]
        a = torch.randn(M + 128, K, device="cuda").half()
        w = (torch.randn(N, taps, K, device="cuda") / (K * taps) ** 0.5).half()
        bias = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda").half() if res else None
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        bn = 128 if N % 128 == 0 else 64
        nblk = (M // 256) * (N // bn)
        tbuf = torch.zeros(nblk * 8 * 8, device="cuda", dtype=torch.int32)
        d = _ffi.ev_conv_gemm_desc()
        d.dtype = 0
        d.A, d.lda, d.W, d.bias = a[64:].data_ptr(), K, w.data_ptr(), bias.data_ptr()
        d.M, d.N, d.K, d.taps, d.dil, d.center = M, N, K, taps, dil, (taps - 1) // 2
        d.out_scale = 1.0
        if pro == "gelu":
            d.act = 2
        elif pro:
            d.pro_lrelu, d.pro_slope, d.act, d.act_slope = 1, 0.1, 3, 0.1
        if r is not None:
            d.res, d.res_dtype, d.ldres = r.data_ptr(), 0, N
        d.out16, d.ldo = out.data_ptr(), N
        d.row_seq = tbuf.data_ptr()
        for _ in range(3):
            lib.ev_op_conv_gemm(C.byref(d), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.ev_op_conv_gemm(C.byref(d), None)
        e1.record()
        torch.cuda.synchronize()
        t = tbuf.cpu().numpy().view(np.uint32).reshape(nblk, 8, 8).astype(np.float64)
        ok = t[:, :, 7] == 0xC0FFEE
        tps = 2 if bn == 64 else 1
        steps = (K // 32) * ((taps + tps - 1) // tps)
        print("%-12s %8.1f us | %d blocks, %d steps per tile, %.0f %% of the waves reported" % (name, e0.elapsed_time(e1) * 1e3, nblk, steps, 100 * ok.mean()))
        for g, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
            m = t[:, sl, :7].mean(axis=(0, 1))
            tot = m.sum()
            print("   %s: total %7.0f cycles (s_memtime ticks) | " % (g, tot) +
                  " | ".join("%s %6.0f (%4.1f %%)" % (n, v, 100 * v / tot) for n, v in zip(NAMES, m)))
            print("   %s  per step: " % (" " * len(g)) + " | ".join("%s %6.0f" % (n, v / steps) for n, v in zip(NAMES[1:6], m[1:6])))
        del a, w, out, r, tbuf


if __name__ == "__main__":
    main()
