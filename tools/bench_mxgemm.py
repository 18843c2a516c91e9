#!/usr/bin/env python3
"""Tuning probe for conv_gemm_mx_kernel (ev_gemm_mx.h) at the stage-0 / stage-1 sizes of BASELINE configs[1]: times the conv1 form (plane set in,
leaky-relu, plane set out) and the conv2 form (plane set + fp32 residual in, fp32 + plane set out) for k = 3 / 7 / 11.

    python tools/bench_mxgemm.py [--c 128] [--rows 2113536] [--ks 3,7,11]

With the ablation build (python emotivoice_amd/csrc/build.py --variant mxabl EV_MX_ABL;  EVHIP_LIB=emotivoice_amd/csrc/libevhip_mxabl.so) each form is
also timed with a one-chunk main loop (what a tile costs outside its K loop) and without its epilogue (ev_conv_gemm_desc.reserved0 bits 4 / 5)."""
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
    ap.add_argument("--c", type=int, default=128)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--ks", default="3,7,11")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--valid-shift", type=int, default=-1, help=">= 0: pass a row_valid table (one byte per 2^shift rows, the engine's frame validity: stage 1 = 6, stage 0 = 3) with "
                    "every 129th 256-row tile all-gap, as in a B = 32 x 1024-frame forward")
    ap.add_argument("--timeline", action="store_true", help="EV_MXT build (EVHIP_LIB=.../libevhip_mxt.so): chip-wide timeline of the blocks' main loops and epilogues")
    ap.add_argument("--dbg", default="", help="comma list of reserved0 low bits to A/B in one run (8 = one block per tile instead of the persistent tile loop)")
    args = ap.parse_args()
    lib = _ffi.lib()
    Cc = args.c
    M = (args.rows or (270532608 // Cc)) // 256 * 256          # rows x channels of a ResBlock tensor of configs[1] (33024 frame rows x 8192 / C)
    R = M + 2 * PAD
    nch = Cc // 128
    g = torch.Generator(device="cuda").manual_seed(1)

    def plane_set():
        h = torch.randn(R, Cc, device="cuda", generator=g).half()
        q = [torch.randint(0, 255, (R, Cc // 2), device="cuda", dtype=torch.uint8, generator=g) for _ in range(2)]
        s = [torch.full((nch, R, 4), 120, device="cuda", dtype=torch.uint8) for _ in range(2)]
        return h, q, s

    xi, xo, xr = plane_set(), plane_set(), plane_set()
    res = torch.randn(M, Cc, device="cuda", generator=g)
    out = torch.empty(M, Cc, device="cuda")
    bias = torch.zeros(Cc, device="cuda")
    valid = None
    if args.valid_shift >= 0:
        valid = torch.ones(M >> args.valid_shift, dtype=torch.uint8, device="cuda")
        per_tile = 256 >> args.valid_shift
        for t in range(128, M // 256, 129):          # the gap tile behind every 128 live ones
            valid[t * per_tile:(t + 1) * per_tile] = 0
    abl_build = "mxabl" in os.environ.get("EVHIP_LIB", "")
    names = {0: "full"}
    if abl_build:
        names.update({1: "one-chunk main loop", 2: "no epilogue", 3: "one chunk, no epilogue", 4: "no plane stores", 8: "hi plane stored only", 16: "no scale bytes", 32: "slab rows from L2 (16 tiles)"})
    dbgs = [int(x) for x in args.dbg.split(",")] if args.dbg else []
    for k in [int(x) for x in args.ks.split(",")]:
        wg = (np.random.default_rng(k).standard_normal((Cc, k, Cc)) / np.sqrt(Cc * k)).astype(np.float32)
        d_hi = torch.from_numpy(wg.astype(np.float16)).cuda()
        d_mx = torch.from_numpy(mxfp4.pack_weight_planes(wg)).cuda()
        for form in ("conv1", "conv2", "conv2pl"):
            for abl, nm in list(names.items()) + [(0, "full (again)")] + [(-1 - v, "dbg %d" % v) for v in dbgs]:          # (the first timing of a weight set runs 5-15 % slow: clocks)
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
                if form == "conv1":
                    d.act, d.act_slope = 3, 0.1
                elif form == "conv2pl":          # the engine's default conv2: residual rebuilt from the plane set conv1 read, plane set out only
                    rh, rq, rs = xr
                    d.res, d.res_dtype, d.ldres, d.res_inv_slope = rh[PAD:].data_ptr(), 3, Cc, 10.0
                    d.res_x4, d.res_xs, d.res_xs_stride = rq[1][PAD:].data_ptr(), rs[1][0, PAD:].data_ptr(), R * 4
                else:
                    d.res, d.res_dtype, d.ldres, d.out32 = res.data_ptr(), 1, Cc, out.data_ptr()
                if valid is not None:
                    d.row_valid, d.valid_shift = valid.data_ptr(), args.valid_shift
                d.reserved0 = (abl << 4) if abl >= 0 else (-1 - abl)
                for _ in range(6):
                    rc = lib.ev_op_conv_gemm(C.byref(d), None)
                    assert rc == 0, rc
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st = torch.cuda.current_stream()
                e0.record(st)
                for _ in range(args.reps):
                    lib.ev_op_conv_gemm(C.byref(d), C.c_void_p(st.cuda_stream))
                e1.record(st)
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / args.reps * 1e3
                if args.timeline and abl == 0:
                    nblk = (M // 256) * (Cc // 128)
                    tb = torch.zeros(nblk * 4 + nblk * 64, device="cuda", dtype=torch.int64)
                    d.row_seq = tb.data_ptr()
                    lib.ev_op_conv_gemm(C.byref(d), C.c_void_p(st.cuda_stream))
                    torch.cuda.synchronize()
                    d.row_seq = None
                    tall = tb.cpu().numpy()
                    t = tall[:nblk * 4].reshape(nblk, 4)
                    te = tall[nblk * 4:].reshape(nblk, 8, 8).astype(np.float64)
                    t0, t1, t2 = (t[:, i].astype(np.float64) for i in range(3))
                    base = t0.min()
                    t0, t1, t2 = (t0 - base) / 100.0, (t1 - base) / 100.0, (t2 - base) / 100.0          # us
                    live = (t2 - t0) > 1.0          # (all-gap tiles skip the K loop)
                    total = t2.max()
                    print("   timeline: %d blocks, launch %.1f us | tile life %.1f us (main %.1f + epilogue %.1f; p10 / p90 of the epilogue %.1f / %.1f)" %
                          (nblk, total, (t2 - t0)[live].mean(), (t1 - t0)[live].mean(), (t2 - t1)[live].mean(), np.percentile((t2 - t1)[live], 10), np.percentile((t2 - t1)[live], 90)))
                    # how many blocks are resident / inside their epilogue over time (1-us bins)
                    bins = np.arange(0, total + 1.0, 1.0)
                    res_n = np.zeros(len(bins)); epi_n = np.zeros(len(bins))
                    for a_, b_, arr in ((t0, t2, res_n), (t1, t2, epi_n)):
                        np.add.at(arr, np.clip(a_.astype(int), 0, len(bins) - 1), 1)
                        np.add.at(arr, np.clip(b_.astype(int), 0, len(bins) - 1), -1)
                    res_c, epi_c = np.cumsum(res_n), np.cumsum(epi_n)
                    mid = slice(int(0.1 * len(bins)), int(0.9 * len(bins)))
                    frac = epi_c[mid] / np.maximum(res_c[mid], 1)
                    print("   resident blocks (mid 80 %% of the launch): mean %.0f | in their epilogue: mean %.0f = %.2f of the resident ones, min %.2f, max %.2f, std %.2f" %
                          (res_c[mid].mean(), epi_c[mid].mean(), frac.mean(), frac.min(), frac.max(), frac.std()))
                    # store drain of wave 0 (last store issued -> all acknowledged) and the turnover of a CU slot: blocks grouped by CU (XCC id + HW_ID's SE / SH / CU bits),
                    # gap = start of a block - end (wave 0, stores acknowledged) of the latest block of that CU that ended before it while the CU was full
                    drain = (t[:, 3] & 0xffffffff).astype(np.float64) / 100.0
                    hw = (t[:, 3] >> 32) & 0xffff
                    xcc = (t[:, 3] >> 48) & 0xf
                    cu = (xcc << 16) | (hw & 0xff00)          # HW_ID: [11:8] cu_id, [12] sh_id, [15:13] se_id
                    gaps = []
                    for c in np.unique(cu):
                        idx = np.nonzero(cu == c)[0]
                        st, en = np.sort(t0[idx]), np.sort(t2[idx])
                        for k_ in range(2, len(st)):          # the k-th start on a 2-slot CU follows the (k - 2)-th end
                            gaps.append(st[k_] - en[k_ - 2])
                    gaps = np.array(gaps)
                    print("   store drain of wave 0: mean %.2f us (p90 %.2f) | CU slot turnover (end of wave 0 -> next block's start): median %.2f us, mean %.2f, p10 %.2f, p90 %.2f over %d hand-overs on %d CUs" %
                          (drain[live].mean(), np.percentile(drain[live], 90), np.median(gaps), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 90), len(gaps), len(np.unique(cu))))
                    # inside the epilogue (s_memtime ticks per wave): setup (bias / row-valid loads, address arithmetic), then the four 16-row passes
                    lv = live[:, None] & (te[:, :, 0] > 0)
                    dd = np.diff(te[:, :, :6], axis=2)
                    print("   epilogue per wave (s_memtime ticks): setup %.0f | passes %s | total %.0f" %
                          (dd[:, :, 0][lv].mean(), " ".join("%.0f" % dd[:, :, 1 + i][lv].mean() for i in range(4)), (te[:, :, 5] - te[:, :, 0])[lv].mean()))
                    # skew of the waves' epilogue ends inside a block (the block's slot is free when its LAST wave ends; the turnover above is counted from wave 0's end)
                    ends = te[:, :, 5]
                    okb = live & (te[:, :, 0] > 0).all(axis=1)
                    tick_us = (t2 - t1)[okb].mean() / np.maximum((te[:, 0, 5] - te[:, 0, 0])[okb].mean(), 1.0)
                    print("   wave skew inside a block: last wave's end - first wave's end %.2f us (p90 %.2f); last wave's end - wave 0's end %.2f us" %
                          ((ends.max(axis=1) - ends.min(axis=1))[okb].mean() * tick_us, np.percentile((ends.max(axis=1) - ends.min(axis=1))[okb], 90) * tick_us,
                           (ends.max(axis=1) - ends[:, 0])[okb].mean() * tick_us))
                    ent = te[:, :, 0]
                    print("   per wave (0..7), relative to the block's first wave: epilogue entry %s us | epilogue end %s us | duration %s us" %
                          (" ".join("%.2f" % v for v in ((ent - ent.min(axis=1, keepdims=True))[okb].mean(axis=0) * tick_us)),
                           " ".join("%.2f" % v for v in ((ends - ends.min(axis=1, keepdims=True))[okb].mean(axis=0) * tick_us)),
                           " ".join("%.2f" % v for v in ((ends - ent)[okb].mean(axis=0) * tick_us))))
                    # first-round skew: start times of the first 512 blocks, and block lifetimes by round
                    order = np.argsort(t0)
                    print("   starts of the first 512 blocks: %.1f .. %.1f us; epilogue length by start-time decile: %s" %
                          (t0[order[0]], t0[order[min(511, nblk - 1)]], " ".join("%.1f" % (t2 - t1)[order[i * nblk // 10:(i + 1) * nblk // 10]].mean() for i in range(10))))
                fl = 2.0 * M * Cc * Cc * k
                by = M * Cc * (3.0625 * 2 + (8 if form == "conv2" else (2.53 if form == "conv2pl" else 0)))
                print("C=%d M=%d k=%2d %-5s %-24s %8.1f us  %6.0f TF/s alg  %5.2f TB/s contract" % (Cc, M, k, form, nm, us, fl / us / 1e6, by / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
