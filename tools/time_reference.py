#!/usr/bin/env python3
"""Time the REFERENCE ITSELF on this container's host cores (cpu_baseline kind "reference").

/root/reference does not exist on the GPU box, so bench.py's live cpu_baseline there is the oracle (kind "port").  This script keeps a
kind-"reference" figure current: it imports the reference in place with tests/golden/make_golden.py's loader, loads the seeded synthetic
checkpoint ("bench" weights: exactly 4 frames per phoneme, the weights bench.py times) into the reference's own JETSGenerator
(load_state_dict strict) and times the call the reference's scripts make (inference_am_vocoder_joint.py:115-131: B = 1, keyword args, no_grad,
then the int16 epilogue) for 64 phonemes (BASELINE configs[0]) and 256 phonemes (configs[1]'s utterance length).  Weight-norm is re-evaluated in
every forward, as in the reference (models/hifigan/models.py:10-14; the joint path never calls remove_weight_norm).

    python tools/time_reference.py [--threads 8] [--seconds 20]   ->  profiles/r5_reference_cpu.json   (bench.py quotes it next to its own numbers)
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r5_reference_cpu.json"))
    args = ap.parse_args()
    if not os.path.isdir("/root/reference"):
        sys.exit("/root/reference is not here: this script only runs in the build container")
    from make_golden import load_reference
    from emotivoice_amd.synthetic import synth_inputs, synth_state_dict
    torch.set_num_threads(args.threads)
    gen = load_reference()
    sd = synth_state_dict(0, "bench")
    gen.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    res = {}
    for ph in (64, 256):
        utts = synth_inputs(1, [ph] * 64, None)

        def call(u):
            ling = torch.from_numpy(u["ling"]).long().unsqueeze(0)
            with torch.no_grad():
                out = gen(inputs_ling=ling, inputs_style_embedding=torch.from_numpy(u["style"]).unsqueeze(0), input_lengths=torch.tensor([ling.shape[1]]),
                          inputs_content_embedding=torch.from_numpy(u["content"]).unsqueeze(0), inputs_speaker=torch.tensor([0]), alpha=1.0)
            wav = (out["wav_predictions"].squeeze() * 32768.0).cpu().numpy().astype("int16")          # inference_am_vocoder_joint.py:130-131
            return int(out["dec_outputs"].shape[1]), wav

        call(utts[0])          # warm-up
        frames, n, per = 0, 0, []
        t0 = time.perf_counter()
        for u in utts[1:]:
            t1 = time.perf_counter()
            f, _ = call(u)
            per.append(time.perf_counter() - t1)
            frames += f; n += 1
            if time.perf_counter() - t0 > args.seconds:
                break
        dt = time.perf_counter() - t0
        res["%dph" % ph] = dict(phonemes=ph, utterances=n, frames=frames, seconds=round(dt, 2), frames_per_s=round(frames / dt, 1),
                                ms_per_utterance_median=round(float(np.median(per)) * 1e3, 1), x_realtime=round(frames / dt * 256 / 16000, 2))
        print(ph, res["%dph" % ph], flush=True)
    cpu = "?"
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    out = dict(kind="reference", what="the reference's own JETSGenerator (jets.py:50-71), imported in place from /root/reference, seeded 'bench' weights, B = 1 loop + int16 epilogue",
               cores=args.threads, host=dict(cpu=cpu, logical_cpus=os.cpu_count(), machine=platform.machine()), torch=torch.__version__,
               note="build container, not the GPU box's host (the reference cannot travel there); bench.py's live cpu_baseline is the oracle ('port') on the GPU box",
               results=res)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
