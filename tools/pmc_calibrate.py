#!/usr/bin/env python3
"""Known-size traffic for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box (MI355X_MICROARCH.md section HBM: on
gfx950 FETCH_SIZE reports half of a wide coalesced read; WRITE_SIZE is uncalibrated).  Three patterns of exactly 1 GiB read +
1 GiB written per dispatch, larger than the 256 MiB Infinity Cache: a device-to-device copy, an fp16 elementwise kernel with
16-byte accesses (the access shape of the engine's epilogues), and an fp32 one.  tools/profile_summary.py divides the counter
values of these dispatches by 2^30 to get the correction factors it applies to the engine's kernels."""
import torch

N = 1 << 30
x = torch.empty(N, dtype=torch.uint8, device="cuda").random_(0, 255)
y = torch.empty_like(x)
h = torch.randn(N // 2, device="cuda", dtype=torch.float16)
h2 = torch.empty_like(h)
f = torch.randn(N // 4, device="cuda", dtype=torch.float32)
f2 = torch.empty_like(f)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)                      # __amd_rocclr_copyBuffer
    torch.mul(h, 0.5, out=h2)       # vectorized elementwise kernel, fp16
    torch.mul(f, 0.5, out=f2)       # vectorized elementwise kernel, fp32
    torch.cuda.synchronize()
print("calibration traffic done: 1 GiB read + 1 GiB written per dispatch")
