#!/usr/bin/env python3
"""Which phases of the forward run power-limited?  Loops one workload at a time for a few seconds while a thread samples rocm-smi (sclk, socket power), and prints the
median clock / power per workload: the whole forward (configs[1]), the acoustic model alone, the generator alone, and single launches of the generator's kernel
families at their configs[1] sizes.

    python tools/power_probe.py [--seconds 3]
"""
import argparse
import ctypes as C
import math
import os
import re
import subprocess
import sys
import threading
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from emotivoice_amd import _ffi, mxfp4  # noqa: E402
from emotivoice_amd.engine import EVEngine  # noqa: E402
from emotivoice_amd.sharding import broadcast_blob  # noqa: E402

PAD = 64
samples = []
stop = False


def sampler():
    while not stop:
        t = time.perf_counter()
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            continue
        m1 = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        m2 = re.search(r"Power \(W\): ([\d.]+)", out)
        if m1 and m2:
            samples.append((t, int(m1.group(1)), float(m2.group(1))))


def run_case(name, fn, seconds):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        n += 4
    t1 = time.perf_counter()
    mine = [s for s in samples if t0 + 0.7 < s[0] < t1 - 0.1]          # (the power reading lags the load by a few hundred ms)
    ms = (t1 - t0) / n * 1e3
    if mine:
        print("%-58s %8.3f ms / call   sclk median %4d MHz (min %4d)   power median %4.0f W   [%d samples]" %
              (name, ms, int(np.median([s[1] for s in mine])), min(s[1] for s in mine), float(np.median([s[2] for s in mine])), len(mine)), flush=True)
    else:
        print("%-58s %8.3f ms / call   (no samples)" % (name, ms), flush=True)
    time.sleep(1.0)


def main():
    global stop
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    a = ap.parse_args()
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    lib = _ffi.lib()
    dev = torch.device("cuda", 0)
    blob = broadcast_blob(0, 1, 0, None, dur_mode="bench")
    eng = EVEngine(device_id=0)
    eng.load_blob_device(blob.data_ptr(), blob.numel(), keepalive=blob)
    args = types.SimpleNamespace(mode="am_vocoder", batch=32, sub_batches=1, phonemes=256)
    w = bench.Workload(args, eng, 0, dev, torch, _ffi)
    run_case("whole forward, configs[1] (mx)", w.step, a.seconds)
    argsv = types.SimpleNamespace(mode="vocoder", batch=32, sub_batches=1, phonemes=256)
    wv = bench.Workload(argsv, eng, 0, dev, torch, _ffi)
    run_case("generator alone, 32 x 1024-frame mels (mx)", wv.step, a.seconds)
    engf = EVEngine(device_id=0, precision="fast")
    engf.load_blob_device(blob.data_ptr(), blob.numel(), keepalive=blob)
    wf = bench.Workload(args, engf, 0, dev, torch, _ffi)
    run_case("whole forward, configs[1] (fast: fp16 operands)", wf.step, a.seconds)

    # ---- single launches of conv_gemm_mx_kernel at the stage-1 size (C = 128, 2.1 M rows): conv1 form (planes in, planes out)
    g = torch.Generator(device="cuda").manual_seed(1)
    Cc = 128
    M = 2113536
    R = M + 2 * PAD

    def plane_set():
        h = torch.randn(R, Cc, device="cuda", generator=g).half()
        q = [torch.randint(0, 255, (R, Cc // 2), device="cuda", dtype=torch.uint8, generator=g) for _ in range(2)]
        s = [torch.full((1, R, 4), 120, device="cuda", dtype=torch.uint8) for _ in range(2)]
        return h, q, s
    xi, xo = plane_set(), plane_set()
    bias = torch.zeros(Cc, device="cuda")
    keep = []
    for k in (3, 11):
        wg = (np.random.default_rng(k).standard_normal((Cc, k, Cc)) / np.sqrt(Cc * k)).astype(np.float32)
        d_hi = torch.from_numpy(wg.astype(np.float16)).cuda()
        d_mx = torch.from_numpy(mxfp4.pack_weight_planes(wg)).cuda()
        d = _ffi.ev_conv_gemm_desc()
        h, q, s = xi
        d.dtype, d.A, d.lda, d.W, d.W_lo, d.W_mx = 3, h[PAD:].data_ptr(), Cc, d_hi.data_ptr(), d_hi.data_ptr(), d_mx.data_ptr()
        d.mx_x4[0], d.mx_x4[1] = q[0][PAD:].data_ptr(), q[1][PAD:].data_ptr()
        d.mx_xs[0], d.mx_xs[1], d.mx_xs_stride = s[0][0, PAD:].data_ptr(), s[1][0, PAD:].data_ptr(), R * 4
        d.bias, d.M, d.N, d.K, d.taps, d.dil, d.center, d.out_scale, d.ldo = bias.data_ptr(), M, Cc, Cc, k, 1, (k - 1) // 2, 1.0, Cc
        oh, oq, osc = xo
        d.mxo_h, d.mxo_logC, d.mxo_slope, d.mxo_qs_stride = oh[PAD:].data_ptr(), 0, 0.1, R * 4
        d.mxo_q4[0], d.mxo_q4[1] = oq[0][PAD:].data_ptr(), oq[1][PAD:].data_ptr()
        d.mxo_qs[0], d.mxo_qs[1] = osc[0][0, PAD:].data_ptr(), osc[1][0, PAD:].data_ptr()
        d.act, d.act_slope = 3, 0.1
        keep.append((d, d_hi, d_mx))
        st = torch.cuda.current_stream()
        run_case("conv_gemm_mx_kernel C = 128, k = %d, conv1 form, 2.1 M rows" % k, lambda d=d: lib.ev_op_conv_gemm(C.byref(d), C.c_void_p(st.cuda_stream)), a.seconds)
    del xi, xo
    torch.cuda.empty_cache()

    # ---- fused C = 32 pair (stage 3: 8.45 M rows, fp32 in / out)
    Cc, M = 32, 33024 * 256
    full = torch.randn(M + 2 * PAD, Cc, device="cuda", generator=g)
    x = full[PAD:PAD + M]
    out = torch.empty(M, Cc, device="cuda")
    for k in (3, 11):
        rng = np.random.default_rng(k)
        wg1 = (rng.standard_normal((Cc, k, Cc)) / math.sqrt(Cc * k)).astype(np.float32)
        wg2 = (rng.standard_normal((Cc, k, Cc)) / math.sqrt(Cc * k)).astype(np.float32)
        w1h, w2h = torch.from_numpy(wg1.astype(np.float16)).cuda(), torch.from_numpy(wg2.astype(np.float16)).cuda()
        w1m, w2m = torch.from_numpy(mxfp4.pack_pair_weight_planes(wg1)).cuda(), torch.from_numpy(mxfp4.pack_pair_weight_planes(wg2)).cuda()
        b1, b2 = torch.randn(Cc, device="cuda") * 0.1, torch.randn(Cc, device="cuda") * 0.1
        d = _ffi.ev_res_pair_desc()
        d.x, d.ldx, d.w1, d.b1, d.w2, d.M, d.k, d.dil = x.data_ptr(), Cc, w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), M, k, 1
        d.w1_mx, d.w2_mx = w1m.data_ptr(), w2m.data_ptr()
        e = d.epi
        e.bias, e.res, e.res_dtype, e.ldres = b2.data_ptr(), x.data_ptr(), 1, Cc
        e.out_scale, e.out32, e.ldo = 1.0 / 3.0, out.data_ptr(), Cc
        keep.append((d, w1h, w2h, w1m, w2m, b1, b2))
        st = torch.cuda.current_stream()
        run_case("fused C = 32 pair, k = %d, 8.45 M rows" % k, lambda d=d: lib.ev_op_resblock_pair_c32_mx(C.byref(d), C.c_void_p(st.cuda_stream)), a.seconds)
    stop = True
    idle = [s for s in samples[-3:]]
    print("samples total %d" % len(samples))


if __name__ == "__main__":
    main()
