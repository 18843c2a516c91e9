#!/usr/bin/env python3
"""Tuning probe for conv_c64_mx_kernel (ev_conv64_mx.h) at the stage-2 size of BASELINE configs[1]: times conv1 / conv2 forms for k = 3 / 7 / 11 with
one ingredient dropped at a time (ev_conv_gemm_desc.reserved0 bits 4-9).    python tools/bench_c64.py [--rows 4227072]
Per-section cycle sums need the timing build:  python emotivoice_amd/csrc/build.py --variant c64t EV_C64_TIMING;  EVHIP_LIB=emotivoice_amd/csrc/libevhip_c64t.so python tools/bench_c64.py"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi, mxfp4  # noqa: E402

PAD = 64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4227072)
    ap.add_argument("--ks", default="3,7,11")
    ap.add_argument("--ab", action="store_true", help="only the A/B of the two kernels that can run a k = 7 / 11 launch: the streamed conv_gemm_mx64_kernel (default) against the persistent conv_c64_mx_kernel (reserved0 bit 3)")
    args = ap.parse_args()
    lib = _ffi.lib()
    M, Cc = args.rows // 256 * 256, 64
    R = M + 2 * PAD
    g = torch.Generator(device="cuda").manual_seed(1)
    h = torch.randn(R, Cc, device="cuda", generator=g).half()
    q = [torch.randint(0, 255, (R, 32), device="cuda", dtype=torch.uint8, generator=g) for _ in range(2)]
    s = [torch.full((R, 4), 120, device="cuda", dtype=torch.uint8) for _ in range(2)]
    oh = torch.empty(R, Cc, device="cuda", dtype=torch.float16)
    oq = [torch.empty(R, 32, device="cuda", dtype=torch.uint8) for _ in range(2)]
    osc = [torch.empty(R, 4, device="cuda", dtype=torch.uint8) for _ in range(2)]
    res = torch.randn(M, Cc, device="cuda", generator=g)
    out = torch.empty(M, Cc, device="cuda")
    bias = torch.zeros(Cc, device="cuda")
    names_ab = {0: "streamed (mx64)", -8: "persistent (c64)", -1: "streamed (again)", -9: "persistent (again)"}
    names = {0: "full", 1: "no plane stores", 2: "no fp32 stores", 4: "no slab requests", 8: "no MFMAs", 16: "no LDS slab writes", 32: "no plane quantisation",
             3: "no stores at all", 63: "loop skeleton only"}
    for k in [int(x) for x in args.ks.split(",")]:
        wg = (np.random.default_rng(k).standard_normal((Cc, k, Cc)) / np.sqrt(Cc * k)).astype(np.float32)
        d_hi = torch.from_numpy(wg.astype(np.float16)).cuda()
        d_mx = torch.from_numpy(mxfp4.pack_c64_weight_planes(wg)).cuda()
        for form in ("conv1", "conv2", "conv2pl") if args.ab else ("conv1", "conv2"):
            for abl, nm in (names_ab if args.ab else names).items():
                d = _ffi.ev_conv_gemm_desc()
                d.dtype, d.A, d.lda, d.W, d.W_lo, d.W_mx = 3, h[PAD:].data_ptr(), Cc, d_hi.data_ptr(), d_hi.data_ptr(), d_mx.data_ptr()
                d.mx_x4[0], d.mx_x4[1], d.mx_xs[0], d.mx_xs[1], d.mx_xs_stride = q[0][PAD:].data_ptr(), q[1][PAD:].data_ptr(), s[0][PAD:].data_ptr(), s[1][PAD:].data_ptr(), R * 4
                d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale, d.ldo = bias.data_ptr(), M, Cc, Cc, k, 1, (k - 1) // 2, 1.0, Cc
                d.mxo_h, d.mxo_logC, d.mxo_slope, d.mxo_qs_stride = oh[PAD:].data_ptr(), 6, 0.1, R * 4
                d.mxo_q4[0], d.mxo_q4[1], d.mxo_qs[0], d.mxo_qs[1] = oq[0][PAD:].data_ptr(), oq[1][PAD:].data_ptr(), osc[0][PAD:].data_ptr(), osc[1][PAD:].data_ptr()
                if form == "conv1":
                    d.act, d.act_slope = 3, 0.1
                elif form == "conv2pl":          # the engine's conv2: residual from the plane set conv1 read, plane set out
                    d.res, d.res_dtype, d.ldres, d.res_inv_slope, d.out_scale = h[PAD:].data_ptr(), 3, Cc, 10.0, 1.0
                    d.res_x4, d.res_xs, d.res_xs_stride = q[1][PAD:].data_ptr(), s[1][PAD:].data_ptr(), R * 4
                else:
                    d.res, d.res_dtype, d.ldres, d.out32 = res.data_ptr(), 1, Cc, out.data_ptr()
                d.reserved0 = (abl << 4) if abl >= 0 and not args.ab else (8 if abl in (-8, -9) else 0)
                for _ in range(2):
                    assert lib.ev_op_conv_gemm(C.byref(d), None) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st = torch.cuda.current_stream()
                e0.record(st)
                for _ in range(5):
                    lib.ev_op_conv_gemm(C.byref(d), C.c_void_p(st.cuda_stream))
                e1.record(st)
                torch.cuda.synchronize()
                print("k=%2d %-5s %-24s %8.1f us" % (k, form, nm, e0.elapsed_time(e1) / 5 * 1e3), flush=True)
            if form == "conv2" and os.environ.get("EVHIP_LIB"):          # tuning build with -DEV_C64_TIMING (EVHIP_LIB=.../libevhip_c64t.so): where a wave's cycles go
                d.reserved0 = 64 << 4
                lib.ev_op_conv_gemm(C.byref(d), None)
                torch.cuda.synchronize()
                tk = out.view(torch.int32)[0, :7].cpu().numpy().astype(np.int64)
                n = max(int(tk[6]), 1)
                print("   per item (100 MHz ticks): requests %d | MFMA phase %d | barrier %d | slab -> LDS %d | epilogue %d | barrier %d   (%d items)"
                      % (tk[0] // n, tk[1] // n, tk[2] // n, tk[3] // n, tk[4] // n, tk[5] // n, n), flush=True)


if __name__ == "__main__":
    main()
