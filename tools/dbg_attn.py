import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emotivoice_amd import _ffi
lib = _ffi.lib()
torch.manual_seed(7)
Cc, H = 384, 8
for lens in ([70], [32], [33], [48], [64], [70, 1, 130, 64, 300, 129, 1024]):
    offs, rows = [], 4
    for n in lens:
        offs.append(rows); rows += n + 4
    qkv = torch.randn(rows, 3 * Cc, device="cuda")
    out = torch.zeros(rows, Cc, device="cuda")
    so = torch.tensor(offs, dtype=torch.int32, device="cuda"); sl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    assert lib.ev_op_attention(qkv.data_ptr(), 2, Cc, H, so.data_ptr(), sl.data_ptr(), len(lens), max(lens), out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    for o, n in zip(offs, lens):
        blk = qkv[o:o + n].double().cpu()
        q, k, v = [t.view(n, H, 48).transpose(0, 1) for t in blk.split(Cc, dim=1)]
        att = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(48), dim=-1) @ v
        ref = att.transpose(0, 1).reshape(n, Cc)
        got = out[o:o + n].double().cpu()
        err = (got - ref).norm() / ref.norm()
        perq = ((got - ref).norm(dim=1) / ref.norm(dim=1))
        perd = ((got - ref).view(n, H, 48).norm(dim=(0, 1)) / ref.view(n, H, 48).norm(dim=(0, 1)))
        print("lens", lens if len(lens) < 3 else "...", "n", n, "err %.2e" % err, "worst queries", [(int(i), "%.1e" % perq[i]) for i in perq.argsort(descending=True)[:4]],
              "per-d max %.1e min %.1e" % (perd.max(), perd.min()), "d argmax", int(perd.argmax()))
