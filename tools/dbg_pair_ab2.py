"""Debug: run-to-run differences of the C = 32 MX pair kernels with accumulate-in (k = 3)."""
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
k, dil = 3, 1
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


def run(dbg, inplace, zero_acc=False):
    a_in = torch.zeros_like(acc) if zero_acc else acc.clone()
    out = a_in if inplace else torch.full((M, Cc), 7.0, device="cuda")
    d = _ffi.ev_res_pair_desc()
    d.x, d.ldx, d.w1, d.b1, d.w2, d.M, d.k, d.dil = x.data_ptr(), Cc, w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), M, k, dil
    d.w1_mx, d.w2_mx = w1m.data_ptr(), w2m.data_ptr()
    e = d.epi
    e.bias, e.res, e.res_dtype, e.ldres = b2.data_ptr(), x.data_ptr(), 1, Cc
    e.row_valid, e.valid_shift, e.out_scale = valid.data_ptr(), 4, 1.0 / 3.0
    e.acc32, e.ldacc = a_in.data_ptr(), Cc
    e.out32, e.ldo = out.data_ptr(), Cc
    e.reserved0 = dbg
    torch.cuda.synchronize()
    assert lib.ev_op_resblock_pair_c32_mx(C.byref(d), None) == 0
    torch.cuda.synchronize()
    return out.cpu()


for name, dbg, inplace, zacc in (("lock in-place", 4, True, False), ("lock separate", 4, False, False), ("lock separate zero acc", 4, False, True),
                                 ("mx2 in-place", 0, True, False), ("mx2 separate", 0, False, False)):
    outs = [run(dbg, inplace, zacc) for _ in range(6)]
    ndiff = [int((outs[0] != o).any(1).sum()) for o in outs[1:]]
    allrows = set()
    for o in outs[1:]:
        allrows |= set(torch.nonzero((outs[0] != o).any(1)).flatten().tolist())
    cols = set()
    for o in outs[1:]:
        cols |= set(torch.nonzero((outs[0] != o).any(0)).flatten().tolist())
    print(name, "rows differing vs run 0:", ndiff, "rows", sorted(allrows)[:24], "cols", sorted(cols)[:32], "max", max(float((outs[0] - o).abs().max()) for o in outs[1:]))
