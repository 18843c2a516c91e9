#!/usr/bin/env python3
"""Calibration of the zero-mean / "_hot" synthetic generator weights (emotivoice_amd/synthetic.py ZDC_POST_BIAS / HOT_*), per WEIGHT SEED: prints the conv_post
bias that gives a zero-mean waveform with the plain weights, and for the "_hot" gains the per-stage activation levels and the conv_post gain / bias that give an
unsaturated (pre-tanh rms ~0.5), zero-mean waveform.  CPU only (the oracle).

    python tools/calibrate_hot.py [--seed 1]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import EVShapes, am_forward, hifigan_forward, synth_inputs, synth_state_dict  # noqa: E402
from oracle.jets_oracle import to_torch_sd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0, help="weight seed (synth_state_dict's first argument)")
    seed = ap.parse_args().seed
    torch.set_num_threads(8)
    shapes = EVShapes()
    utts = synth_inputs(21, [48, 64], [7, 1234])
    # plain weights: the conv_post bias of a zero-mean waveform
    sdz = synth_state_dict(seed, "parity")
    sdzt = to_torch_sd(sdz)
    with torch.no_grad():
        melz = []
        for u in utts:
            am = am_forward(sdzt, torch.from_numpy(u["ling"]), u["speaker"], torch.from_numpy(u["style"]), torch.from_numpy(u["content"]), shapes)
            melz.append(am["dec_outputs"].t().contiguous())
        lo, hi = -20.0, 20.0
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            sdz["generator.conv_post.bias"][...] = mid
            sdzt = to_torch_sd(sdz)
            if np.mean([float(hifigan_forward(sdzt, mel, shapes).mean()) for mel in melz]) > 0:
                hi = mid
            else:
                lo = mid
        for mel in melz:
            wv = hifigan_forward(sdzt, mel, shapes)
            print("plain weights: wav mean %.4f std %.4f max %.3f" % (float(wv.mean()), float(wv.std()), float(wv.abs().max())))
        print("ZDC_POST_BIAS[%d] = %.4f" % (seed, 0.5 * (lo + hi)))
    sdn = synth_state_dict(seed, "parity_zdc_hot", post_gain=1.0)
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
        sdn2 = synth_state_dict(seed, "parity_zdc_hot", post_gain=gain)
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
        print("HOT_POST_GAIN[%d] = %.6g\nHOT_POST_BIAS[%d] = %.6g" % (seed, gain, seed, 0.5 * (lo + hi)))


if __name__ == "__main__":
    main()
