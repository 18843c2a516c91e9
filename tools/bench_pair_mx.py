#!/usr/bin/env python3
"""Times the fused C = 32 ResBlock pair on stage 3 of the B = 32 x 256-phoneme forward (8.45 M rows x 32 channels, fp32 in / out) with in-process A/B over
ev_res_pair_desc.epi.reserved0: 0 = the launcher's default (round 6: resblock_pair_c32_e5_kernel, E5M2 activation operands, ev_pair_e5.h, at k = 3; the fp4 kernel at k = 7 / 11), 32 = E5M2 at every k; 16 = block-scaled fp4
activation operands (resblock_pair_c32_mx2_kernel, ev_pair_mx.h: rounds 3-5), 16 + 4 = its lock-step form, 16 + 8 = its round-4 instruction stream (the kernel's
r4_paths switch).  Outputs of one format are compared bit for bit with the first run of that format (the two formats differ in the cross terms' last bits).

    python tools/bench_pair_mx.py [--ks 3,7,11] [--dbg 0,8,0,8,4]"""
import argparse
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi, mxfp4  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="3")
    ap.add_argument("--dils", default="1,3,5")
    ap.add_argument("--dbg", default="0,16,0,16")
    ap.add_argument("--rows", type=int, default=33024 * 256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--timing", action="store_true", help="EV_PAIR_TIMING build (EVHIP_LIB=.../libevhip_ptime.so): print where a wave's cycles go (default format only)")
    a = ap.parse_args()
    lib = _ffi.lib()
    Cc, M, PAD = 32, a.rows, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    full = torch.randn(M + 2 * PAD, Cc, device="cuda", generator=g)
    full[:PAD] = 0
    full[PAD + M:] = 0
    x = full[PAD:PAD + M]
    valid = torch.ones(M // 256, dtype=torch.uint8, device="cuda")          # the engine's frame validity at stage 3 (shift 8); a gap frame every 1028
    valid[1024::1028] = 0
    acc = torch.randn(M, Cc, device="cuda", generator=g)
    for k in [int(v) for v in a.ks.split(",")]:
        rng = np.random.default_rng(k)
        wg1 = (rng.standard_normal((Cc, k, Cc)) / math.sqrt(Cc * k)).astype(np.float32)
        wg2 = (rng.standard_normal((Cc, k, Cc)) / math.sqrt(Cc * k)).astype(np.float32)
        w1h, w2h = torch.from_numpy(wg1.astype(np.float16)).cuda(), torch.from_numpy(wg2.astype(np.float16)).cuda()
        w1m, w2m = torch.from_numpy(mxfp4.pack_pair_weight_planes(wg1)).cuda(), torch.from_numpy(mxfp4.pack_pair_weight_planes(wg2)).cuda()
        b1, b2 = torch.randn(Cc, device="cuda") * 0.1, torch.randn(Cc, device="cuda") * 0.1
        for dil in [int(v) for v in a.dils.split(",")]:
            for acc_in in (False, True):
                first = None
                for dbg in [int(v) for v in a.dbg.split(",")]:
                    out = acc.clone() if acc_in else torch.empty(M, Cc, device="cuda")
                    d = _ffi.ev_res_pair_desc()
                    d.x, d.ldx, d.w1, d.b1, d.w2, d.M, d.k, d.dil = x.data_ptr(), Cc, w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), M, k, dil
                    d.w1_mx, d.w2_mx = w1m.data_ptr(), w2m.data_ptr()
                    e = d.epi
                    e.bias, e.res, e.res_dtype, e.ldres = b2.data_ptr(), x.data_ptr(), 1, Cc
                    e.row_valid, e.valid_shift, e.out_scale = valid.data_ptr(), 8, 1.0 / 3.0
                    e.out32, e.ldo = out.data_ptr(), Cc
                    if acc_in:
                        e.acc32, e.ldacc = acc.data_ptr(), Cc          # (not in place here: repeated launches must see the same addend)
                    e.reserved0 = dbg
                    tbuf = torch.zeros(256 * 8 * 16, device="cuda", dtype=torch.int32) if a.timing else None
                    if a.timing:
                        e.row_seq = tbuf.data_ptr()
                    for _ in range(2):
                        assert lib.ev_op_resblock_pair_c32_mx(C.byref(d), None) == 0
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        lib.ev_op_resblock_pair_c32_mx(C.byref(d), None)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / a.iters
                    if a.timing and (dbg & 32):
                        tbuf.zero_()
                        lib.ev_op_resblock_pair_c32_mx(C.byref(d), None)
                        torch.cuda.synchronize()
                        t = tbuf.cpu().numpy().view(np.uint32).reshape(256, 8, 16).astype(np.float64)
                        ok = t[:, :, 15] == 0xC0FFEE
                        names = ["loads issue", "conv1", "xt quant", "barrier A", "slab->LDS", "conv2", "epilogue", "barrier B", "prologue"]
                        bmo = 128 - (k - 1)
                        iters = math.ceil(math.ceil(M / bmo) / 512)
                        for gname, sl in (("group 0", slice(0, 4)), ("group 1", slice(4, 8))):
                            m = t[:, sl, :9][ok[:, sl]].mean(axis=0)
                            print("   %s (%d waves, %d iterations): per iteration " % (gname, int(ok[:, sl].sum()), iters) +
                                  " | ".join("%s %.0f" % (n, v / iters) for n, v in zip(names[:8], m[:8])) + " | total %.0f ticks; prologue %.0f" % (m[:8].sum() / iters, m[8]))
                    if first is None:
                        first = {}
                    fk = 'e5' if (dbg & 32) or (k == 3 and not (dbg & 16)) else 'fp4'
                    if fk not in first:
                        first[fk], same = out.clone(), ""
                    else:
                        nbad = int((out.view(torch.int32) != first[fk].view(torch.int32)).sum())
                        same = "  bits == first" if nbad == 0 else "  %d ELEMENTS DIFFER" % nbad
                    fl = 2.0 * 2.0 * M * Cc * Cc * k
                    print("k=%2d dil=%d acc_in=%d dbg=%d  %8.1f us  %6.1f TF/s alg  %5.2f TB/s (x in + out%s)%s" %
                          (k, dil, acc_in, dbg, ms * 1e3, fl / ms / 1e9, (2 + acc_in) * M * Cc * 4 / ms / 1e9, " + acc" if acc_in else "", same), flush=True)


if __name__ == "__main__":
    main()
