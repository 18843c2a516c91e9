#!/usr/bin/env python3
"""Per-wave phase timeline of the conv-GEMM kernel (tuning tool, needs the EV_TRACE build variant):

    python emotivoice_amd/csrc/build.py --variant trace EV_TRACE
    EVHIP_LIB=emotivoice_amd/csrc/libevhip_trace.so python tools/trace_gemm.py --shapes s1_k7,s2_k3 --out gpurun_out/trace

Every wave stamps s_memtime before/after each barrier; sampled blocks dump their stamps.  The report gives, per shape:
prologue (first load -> first barrier release), per-step work phase (barrier release -> next barrier arrival), per-step
barrier wait, epilogue, and what fraction of the wave's lifetime the 16x16x32 MFMAs would need at 16 cycles each.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi  # noqa: E402
from bench_gemm import SHAPES  # noqa: E402

TRACE_N = 192
NBLK = 128


def tile_of(dtype, M, K, N, taps):
    es = 2 if dtype == 0 else 4
    if N % 128 == 0:
        big = (M // 256) * (N // 128) >= (2048 if es == 4 else 256)
        return (256, 128, 2, 2) if big else (128, 128, 2, 2)
    if N % 64 == 0:
        return (256, 64, 4, 1)
    return (256, 32, 4, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="s1_k7,s1_k3,s1_k11,s2_k3,s2_k11,s0_k11,s0_k3")
    ap.add_argument("--out", default="gpurun_out/trace")
    ap.add_argument("--flags", type=int, default=0, help="bit 0: drop the output stores (compute + loads only)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    lib = _ffi.lib()
    lib.ev_trace_set.argtypes = [C.c_void_p]
    lib.ev_trace_set.restype = None
    lib.ev_trace_flags.argtypes = [C.c_int]
    lib.ev_trace_flags.restype = None
    lib.ev_trace_flags(args.flags)
    for name in args.shapes.split(","):
        dtype, M, K, N, taps, dil, res, pro = SHAPES[name]
        tdt = torch.float16 if dtype == 0 else torch.float32
        a = torch.randn(M + 128, K, device="cuda", dtype=torch.float32).to(tdt)
        w = (torch.randn(N, taps, K, device="cuda") / (K * taps) ** 0.5).to(tdt)
        bias = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda").to(tdt) if res else None
        out = torch.empty(M, N, device="cuda", dtype=tdt)
        d = _ffi.ev_conv_gemm_desc()
        d.dtype = dtype
        d.A, d.lda, d.W, d.bias = a[64:].data_ptr(), K, w.data_ptr(), bias.data_ptr()
        d.M, d.N, d.K, d.taps, d.dil, d.center = M, N, K, taps, dil, (taps - 1) // 2
        d.out_scale = 1.0
        if pro:
            d.pro_lrelu, d.pro_slope, d.act, d.act_slope = 1, 0.1, 3, 0.1
        if r is not None:
            d.res, d.res_dtype, d.ldres = r.data_ptr(), dtype, N
        if dtype == 0:
            d.out16 = out.data_ptr()
        else:
            d.out32 = out.data_ptr()
        d.ldo = N
        tr = torch.zeros(NBLK * 4 * (TRACE_N + 8), device="cuda", dtype=torch.int32)
        lib.ev_trace_set(None)
        for _ in range(2):
            lib.ev_op_conv_gemm(C.byref(d), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.ev_trace_set(C.c_void_p(tr.data_ptr()))
        e0.record()
        lib.ev_op_conv_gemm(C.byref(d), None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        t = tr.cpu().numpy().view(np.uint32).reshape(NBLK * 4, TRACE_N + 8)
        np.save(os.path.join(args.out, name + ".npy"), t)
        report(name, t, (dtype, M, K, N, taps), us)
        del a, w, out, r, tr


def report(name, t, shape, us):
    dtype, M, K, N, taps = shape
    BM, BN, WT, WC = tile_of(dtype, M, K, N, taps)
    es = 2 if dtype == 0 else 4
    steps = (K * es // 64) * taps
    mfma_per_step = (BM // WT // 16) * (BN // WC // 16) * (1 if es == 2 else 4)
    ok = t[(t[:, 0] & 0xFFFFF000) == 0xE7ACE000]
    if len(ok) == 0:
        print(name, "no trace records")
        return
    n = int(ok[0, 2])
    st = ok[:, 8:8 + min(n, TRACE_N)].astype(np.int64)
    dt = np.diff(st, axis=1) & 0xFFFFFFFF
    # stamps: 0 start | 1 before first barrier | 2 after | per step: before, after | end
    pro = dt[:, 0]
    bar0 = dt[:, 1]
    work = dt[:, 2:2 + 2 * steps:2]
    wait = dt[:, 3:3 + 2 * steps:2]
    epi = dt[:, 2 + 2 * steps:].sum(axis=1) if dt.shape[1] > 2 + 2 * steps else np.zeros(len(ok))
    total = (st[:, min(n, TRACE_N) - 1] - st[:, 0]) & 0xFFFFFFFF
    ideal = steps * mfma_per_step * (16 if es == 2 else 32)
    print("%-12s tile %dx%d steps %3d  %8.1f us kernel | waves traced %d | cycles per wave: total %7.0f  prologue %6.0f (+bar %5.0f)  "
          "work/step %6.1f (ideal mfma %d)  wait/step %6.1f  epilogue %6.0f | mfma-ideal/total %.3f" %
          (name, BM, BN, steps, us, len(ok), total.mean(), pro.mean(), bar0.mean(), work.mean(), mfma_per_step * (16 if es == 2 else 32),
           wait.mean(), epi.mean(), ideal / total.mean()), flush=True)
    # per-step profile (mean over waves) for the first chunk of steps
    k = min(steps, 3 * taps)
    print("   work[0:%d] " % k + " ".join("%4.0f" % v for v in work[:, :k].mean(axis=0)))
    print("   wait[0:%d] " % k + " ".join("%4.0f" % v for v in wait[:, :k].mean(axis=0)))
    if dt.shape[1] > 3 + 2 * steps:      # stamps inside the epilogue: setup | per pass: transpose-in, then one per iteration
        print("   epilogue segments: " + " ".join("%4.0f" % v for v in dt[:, 2 + 2 * steps:].mean(axis=0)))
    # block phase relations: start / epilogue start / end of wave 0 of every sampled block, in k-cycles from the first start
    # (do the tiles of one dispatch round run in lockstep, i.e. do all epilogues hit HBM at the same time?)
    w0 = (ok[:, 0] & 0xFFF) == 0
    if w0.any() and dt.shape[1] > 2 + 2 * steps:
        s0 = st[w0]
        base = s0[:, 0].min()
        rows = sorted(((int(r[0] - base) & 0xFFFFFFFF) / 1e3, (int(r[2 + 2 * steps] - base) & 0xFFFFFFFF) / 1e3,
                       (int(r[min(n, TRACE_N) - 1] - base) & 0xFFFFFFFF) / 1e3, int(b)) for r, b in zip(s0, ok[w0][:, 1]))
        print("   blocks (start, epilogue start, end) k-cycles: " + " | ".join("%d: %.0f %.0f %.0f" % (b, a_, e_, z_) for a_, e_, z_, b in rows[:40]))
    # co-residency: group traced waves by (xcc, cu, simd) from HW_ID
    hw = ok[:, 3]
    print("   hw_id sample: " + " ".join("%08x" % v for v in hw[:4]) + "  xcc " + " ".join(str(v & 15) for v in ok[:4, 4]))


if __name__ == "__main__":
    main()
