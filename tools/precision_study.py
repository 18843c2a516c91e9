#!/usr/bin/env python3
"""CPU emulation of the engine's rounding points in the HiFi-GAN generator (tuning tool, not part of the product).

Answers "which precision recipe meets 1e-3 on a DC-free waveform": the fp32 oracle forward is re-run with, per generator stage,
  w : 0 = fp32 weights, 1 = fp16 weights, 2 = hi/lo split weights (w_hi + 2^-11 w_lo, error 2^-22)
  a : 0 = fp32 MFMA operands, 1 = fp16 operands, 2 = hi/lo split operands
  s : 0 = fp32 storage of the stage's intermediates, 1 = fp16 storage (what the engine does today)
and reports rel-L2 / DC-free rel-L2 of the waveform against the plain fp32 oracle.

    python tools/precision_study.py [--phonemes 48] [--zero-dc]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import EVShapes, am_forward, hifigan_forward, synth_inputs, synth_state_dict  # noqa: E402
from oracle.jets_oracle import fold_weight_norm, to_torch_sd  # noqa: E402


def r16(x):
    return x.half().float()


def split16(x):
    hi = x.half().float()
    lo = ((x - hi) * 2048.0).half().float() / 2048.0
    return hi + lo


def q(x, mode):
    return x if mode == 0 else (r16(x) if mode == 1 else split16(x))


def emulate(sd, mel_ct, shapes, recipe, post16=True, prefix="generator"):
    """recipe: list of 5 dicts (conv_pre + 4 stages) with keys w, a, s."""
    def conv(x, name, rc, **kw):
        w = q(fold_weight_norm(sd, name), rc["w"])
        return F.conv1d(q(x, rc["a"]), w, sd[name + ".bias"], **kw)

    x = mel_ct.unsqueeze(0)
    rc = recipe[0]
    x = conv(x, prefix + ".conv_pre", rc, padding=3)
    nk = len(shapes.rb_kernels)
    for i, (u, k) in enumerate(zip(shapes.up_rates, shapes.up_kernels)):
        rc_prev, rc = rc, recipe[i + 1]
        x = q(F.leaky_relu(x, 0.1), rc_prev["s"])                 # stored post-lrelu by the producer
        w = q(fold_weight_norm(sd, f"{prefix}.ups.{i}"), rc["w"])
        x = F.conv_transpose1d(q(x, rc["a"]), w, sd[f"{prefix}.ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        x = q(x, rc["s"])
        xs = None
        for j, (rk, dils) in enumerate(zip(shapes.rb_kernels, shapes.rb_dils)):
            r = f"{prefix}.resblocks.{i * nk + j}"
            y = x
            for d_i, d in enumerate(dils):
                xt = conv(F.leaky_relu(y, 0.1), f"{r}.convs1.{d_i}", rc, dilation=d, padding=(rk * d - d) // 2)
                xt = q(F.leaky_relu(xt, 0.1), rc["s"])
                xt = conv(xt, f"{r}.convs2.{d_i}", rc, dilation=1, padding=(rk - 1) // 2)
                y = xt + y
                if d_i + 1 < len(dils):
                    y = q(y, rc["s"])
            y = y / nk
            if j + 1 < nk:
                y = q(y, rc["s"] if rc.get("mrf16", 1) else 0)
            xs = y if xs is None else xs + y
        x = xs
    x = q(F.leaky_relu(x), recipe[-1]["s"])
    x = F.conv1d(x, fold_weight_norm(sd, prefix + ".conv_post"), sd[prefix + ".conv_post.bias"], padding=3)
    return torch.tanh(x).view(-1)


def errs(a, b):
    a, b = a.double().numpy(), b.double().numpy()
    d = np.linalg.norm(a - b)
    return d / np.linalg.norm(b), d / np.linalg.norm(b - b.mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phonemes", type=int, default=48)
    ap.add_argument("--zero-dc", action="store_true")
    ap.add_argument("--seed", type=int, default=21)
    args = ap.parse_args()
    torch.set_num_threads(8)
    shapes = EVShapes()
    sd = to_torch_sd(synth_state_dict(0, "parity"))
    utt = synth_inputs(args.seed, [args.phonemes], [7])[0]
    with torch.no_grad():
        am = am_forward(sd, torch.from_numpy(utt["ling"]), 7, torch.from_numpy(utt["style"]), torch.from_numpy(utt["content"]), shapes)
        mel = am["dec_outputs"].t().contiguous()
        if args.zero_dc:
            # choose conv_post's bias so that the reference waveform has zero mean (bisection on the scalar bias)
            lo, hi = -20.0, 20.0
            for _ in range(30):
                mid = 0.5 * (lo + hi)
                sd["generator.conv_post.bias"] = torch.tensor([mid])
                m = hifigan_forward(sd, mel, shapes).mean().item()
                if m > 0:
                    hi = mid
                else:
                    lo = mid
            print("zero-DC conv_post bias = %.6f" % sd["generator.conv_post.bias"].item())
        ref = hifigan_forward(sd, mel, shapes)
        print("frames %d  wav mean %.4f  std %.4f  |max| %.4f" % (mel.shape[1], ref.mean().item(), ref.std().item(), ref.abs().max().item()))
        F16 = dict(w=1, a=1, s=1)
        EX = dict(w=0, a=0, s=0)
        cases = {
            "engine today (all fp16)": [F16] * 5,
            "fp16 weights only": [dict(w=1, a=0, s=0)] * 5,
            "fp16 operands+storage only": [dict(w=0, a=1, s=1)] * 5,
            "split weights, fp16 operands": [dict(w=2, a=1, s=1)] * 5,
            "fp16 weights, split operands (fp32 storage)": [dict(w=1, a=2, s=0)] * 5,
            "exact pre+st0 only": [EX, EX, F16, F16, F16],
            "exact st1 only": [F16, F16, EX, F16, F16],
            "exact st2 only": [F16, F16, F16, EX, F16],
            "exact st3 only": [F16, F16, F16, F16, EX],
            "exact st2+st3": [F16, F16, F16, EX, EX],
            "exact st2+st3, split-w st0+st1": [dict(w=2, a=1, s=1)] * 3 + [EX, EX],
            "fp16 everywhere but fp32 MRF branches": [dict(w=1, a=1, s=1, mrf16=0)] * 5,
            "fp16 weights + operands, fp32 storage (1 MFMA)": [dict(w=1, a=1, s=0)] * 5,
            "split weights, fp16 operands, fp32 storage (2 MFMA)": [dict(w=2, a=1, s=0)] * 5,
            "fp32 weights, fp16 operands, fp32 storage": [dict(w=0, a=1, s=0)] * 5,
            "fp16 weights, fp32 operands, fp32 storage": [dict(w=1, a=0, s=0)] * 5,
            "1 MFMA + fp32 storage in st0-st1, all fp16 in st2-st3": [dict(w=1, a=1, s=0)] * 3 + [F16, F16],
            "all fp16 in st0-st1, 1 MFMA + fp32 storage in st2-st3": [F16] * 3 + [dict(w=1, a=1, s=0)] * 2,
        }
        for name, rec in cases.items():
            out = emulate(sd, r16(mel), shapes, rec)
            e, eac = errs(out, ref)
            print("%-48s wav %.3e   wav_ac %.3e" % (name, e, eac), flush=True)


if __name__ == "__main__":
    main()
