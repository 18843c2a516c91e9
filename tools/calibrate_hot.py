#!/usr/bin/env python3
"""Calibration of the "_hot" synthetic generator weights (emotivoice_amd/synthetic.py HOT_*): prints the per-stage activation levels and
the conv_post gain / bias that give an unsaturated (pre-tanh rms ~0.5), zero-mean waveform.  CPU only (the oracle)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import EVShapes, am_forward, hifigan_forward, synth_inputs, synth_state_dict  # noqa: E402
from oracle.jets_oracle import to_torch_sd  # noqa: E402


def main():
    torch.set_num_threads(8)
    shapes = EVShapes()
    utts = synth_inputs(21, [48, 64], [7, 1234])
    sdn = synth_state_dict(0, "parity_zdc_hot", post_gain=1.0)
    sdn["generator.conv_post.bias"][...] = 0.0
    sd = to_torch_sd(sdn)
    mels = []
    with torch.no_grad():
        for u in utts:
            am = am_forward(sd, torch.from_numpy(u["ling"]), u["speaker"], torch.from_numpy(u["style"]), torch.from_numpy(u["content"]), shapes)
            mels.append(am["dec_outputs"].t().contiguous())
        taps = {}
        hifigan_forward(sd, mels[0], shapes, taps=taps)
        for k, v in taps.items():
            if k.startswith("voc"):
                print("  %-10s rms %10.3f  max %10.2f" % (k, v.pow(2).mean().sqrt(), v.abs().max()))
        # pre-tanh signal = atanh(wav) is awkward when saturated: recompute conv_post by hand on the last tap
        x = torch.nn.functional.leaky_relu(taps["voc_mrf3"], 0.01).unsqueeze(0)
        from oracle.jets_oracle import fold_weight_norm
        w = fold_weight_norm(sd, "generator.conv_post")
        pre = torch.nn.functional.conv1d(x, w, None, padding=3).reshape(-1)
        gain = 0.5 / float(pre.std())
        print("pre-tanh std at post_gain 1: %.4f -> HOT_POST_GAIN = %.6g" % (float(pre.std()), gain))
        sdn2 = synth_state_dict(0, "parity_zdc_hot", post_gain=gain)
        lo, hi = -5.0, 5.0
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            sdn2["generator.conv_post.bias"][...] = mid
            sd2 = to_torch_sd(sdn2)
            m = np.mean([float(hifigan_forward(sd2, mel, shapes).mean()) for mel in mels])
            if m > 0:
                hi = mid
            else:
                lo = mid
        sdn2["generator.conv_post.bias"][...] = 0.5 * (lo + hi)
        sd2 = to_torch_sd(sdn2)
        for mel in mels:
            wv = hifigan_forward(sd2, mel, shapes)
            print("wav mean %.4f std %.4f max %.3f" % (float(wv.mean()), float(wv.std()), float(wv.abs().max())))
        print("HOT_POST_GAIN = %.6g\nHOT_POST_BIAS = %.6g" % (gain, 0.5 * (lo + hi)))


if __name__ == "__main__":
    main()
