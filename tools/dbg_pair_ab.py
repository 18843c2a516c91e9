"""Debug: resblock_pair_c32_mx2_kernel vs the lock-step kernel on the op test's failing case (k = 3, dil = 5, accumulate-in): where do they differ?"""
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi, mxfp4  # noqa: E402

PAD = 64
lib = _ffi.lib()
for (k, dil, acc_in) in ((3, 5, True), (3, 5, False), (3, 1, True), (3, 3, True), (7, 5, True)):
    torch.manual_seed(300 + k + dil)
    Cc, M = 32, 5 * 256
    full = torch.randn(M + 2 * PAD, Cc, device="cuda") * torch.exp(0.7 * torch.randn(M + 2 * PAD, 1, device="cuda"))
    valid = torch.ones(M // 16, dtype=torch.uint8, device="cuda")
    valid[:2] = 0; valid[30:34] = 0; valid[-3:] = 0
    vrow = valid.repeat_interleave(16).bool()
    full[:PAD] = 0; full[PAD + M:] = 0
    x = full[PAD:PAD + M]
    x[~vrow] = 0
    w1 = torch.randn(Cc, Cc, k, device="cuda") / math.sqrt(Cc * k)
    w2 = torch.randn(Cc, Cc, k, device="cuda") / math.sqrt(Cc * k)
    b1, b2 = torch.randn(Cc, device="cuda") * 0.1, torch.randn(Cc, device="cuda") * 0.1
    acc = torch.randn(M, Cc, device="cuda")

    def wparts(w):
        wg = w.permute(0, 2, 1).contiguous().cpu().numpy()
        return torch.from_numpy(wg.astype(np.float16)).cuda(), torch.from_numpy(mxfp4.pack_pair_weight_planes(wg)).cuda()
    w1h, w1m = wparts(w1)
    w2h, w2m = wparts(w2)
    outs = []
    for dbg in (0, 4, 0, 4):
        out = acc.clone() if acc_in else torch.full((M, Cc), 7.0, device="cuda")
        d = _ffi.ev_res_pair_desc()
        d.x, d.ldx, d.w1, d.b1, d.w2, d.M, d.k, d.dil = x.data_ptr(), Cc, w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), M, k, dil
        d.w1_mx, d.w2_mx = w1m.data_ptr(), w2m.data_ptr()
        e = d.epi
        e.bias, e.res, e.res_dtype, e.ldres = b2.data_ptr(), x.data_ptr(), 1, Cc
        e.row_valid, e.valid_shift, e.out_scale = valid.data_ptr(), 4, 1.0 / 3.0
        if acc_in:
            e.acc32, e.ldacc = out.data_ptr(), Cc
        e.out32, e.ldo = out.data_ptr(), Cc
        e.reserved0 = dbg
        torch.cuda.synchronize()
        assert lib.ev_op_resblock_pair_c32_mx(C.byref(d), None) == 0
        torch.cuda.synchronize()
        outs.append(out.cpu())
    a, b, a2, b2_ = outs
    print("case", (k, dil, acc_in), "mx2 repeatable", bool(torch.equal(a, a2)), "lockstep repeatable", bool(torch.equal(b, b2_)), "equal", bool(torch.equal(a, b)))
    if not torch.equal(a, b):
        diff = (a != b)
        rows = torch.nonzero(diff.any(1)).flatten()
        print("  differing rows:", rows.numel(), rows[:40].tolist(), "max abs", float((a - b).abs().max()), "cols of first row", torch.nonzero(diff[rows[0]]).flatten().tolist()[:16])
        r = int(rows[0])
        print("  row", r, "mx2", a[r, :4].tolist(), "lock", b[r, :4].tolist(), "rows mod 126:", sorted(set((rows % 126).tolist()))[:20], "mod 254:", sorted(set((rows % 254).tolist()))[:20])
