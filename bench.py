#!/usr/bin/env python3
"""Throughput of the EmotiVoice hot path (JETSGenerator.forward = acoustic model + HiFi-GAN) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode am_vocoder|ragged|vocoder|style|pipeline] [--precision mx|fast|strict]
        N > 1: one rank per GPU.  Started by a launcher (torch.distributed.run: WORLD_SIZE set) it is a rank and requires
        WORLD_SIZE == N; started bare it becomes the launcher itself (spawn_command) -- `python bench.py --gpus 8` runs 8 ranks.

Workloads (BASELINE.json configs; seeded synthetic weights and inputs, nothing is read from disk):
  am_vocoder (default) -- configs[1]: batch 32 x 256 synthetic phonemes, one speaker, AM + vocoder end to end; the duration head
              of the "bench" weights gives exactly 4 frames / phoneme (1024 frames = 16.384 s of audio per utterance).
              N > 1 runs configs[3]: 256 utterances per GPU per step as 8 sub-batches of 32 (2048 over 8 GPUs).
  ragged   -- configs[2]: batch 256, lengths 64 + (i * 7919 mod 449), speakers i mod 2000, "parity" weights (predicted durations
              vary: the length regulator's ragged path).
  vocoder  -- configs[4]: 128 pre-computed 80 x 1024 fp16 mels per GPU through ev_vocoder (1024 over 8 GPUs).
  style    -- (not a BASELINE config) the SimBERT prompt / content encoder on the device: texts/s, CPU oracle beside it.
One "step" = one pass of the rank's batch (all its sub-batches) with the inputs already resident in HBM.  N > 1: utterances are
sharded (weak scaling, per-GPU work fixed); the only collective is the start-up broadcast of the packed weight blob over RCCL.
Precision (--precision): "mx" (default, the contract mode: waveform within 1e-3 of the reference on every fixture) = fp32-class activations,
a product as one fp16 MFMA + two block-scaled fp4 MFMAs; "fast" = fp16 MFMA operands / activations (the precision BASELINE.json names, 2.4e-3
on zero-mean audio); "strict" = split precision (three fp16 MFMAs per product, ~1e-6).  At N = 1 the other two precisions are timed as well
and reported under "other_precision".  --mode pipeline times the host pipeline (text line -> int16 wav).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

# the host driver of the GPU boxes only supports dmabuf IPC: without this RCCL's intra-node transport fails with `hipIpcGetMemHandle: invalid argument`
# (it is exported on the boxes already; a launcher that builds its own environment must not lose it -- set before torch / HIP load)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per mel frame (SURVEY.md section 8(d) / BASELINE.md section 3)
VOC_CONV_FLOP_PER_FRAME = 614.105e6 - 0.115e6      # every Conv1d/ConvTranspose1d of the generator except conv_post
VOC_STAGE3_RB_FLOP_PER_FRAME = (9.437 + 22.020 + 34.603) * 1e6   # stage-3 ResBlocks (C = 32): run by the fused pair kernel (fast mode)
VOC_STAGE2_K3_RB_FLOP_PER_FRAME = 18.874e6         # stage-2 k = 3 ResBlock (C = 64): run by the C = 64 fused pair kernel (fast mode)
VOC_BYTES_PER_FRAME = 2.026e6                      # layer-wise fp16 contract
# the mx mode's own layer-wise contract: plane sets of 3.0625 B per element between the >= 64-channel layers (conv1 of a pair:
# planes in / planes out; conv2: planes + fp32 residual in, fp32 + planes out), fp32 running MRF sums, the fused C = 32 pairs at 8 B per element
def mx_contract_bytes_per_frame(residual_from_planes=True, mrf_partial_planes=True):
    """Layer-wise HBM bytes per mel frame of the generator's OWN data flow in the mx mode (not SURVEY 8(d)'s numerator: that is VOC_BYTES_PER_FRAME,
    reported as hbm_algorithmic_*).  P = 3.0625 B per element of a plane set (fp16 hi 2 + two fp4 code planes 0.5 + 0.5 + two E8M0 scales 1/32 each);
    a residual rebuilt from planes reads hi + remainder codes + their scales = 2.53125 B; stages 0-2: every conv planes in / planes out, conv2
    + residual, the last conv of a ResBlock + the running MRF sum (stages 0-1 since round 4, stage 2 since round 6: a partial plane set of R bytes per element, the stage
    output only as the next up-conv's planes); stage 3: fused pairs, fp32 in / out (8 B per element and pair)."""
    P, R = 3.0625, 2.53125
    E, Ein = [2048, 8192, 8192, 8192], [512, 2048, 8192, 8192]       # elements per mel frame of a stage's tensors / of its up-conv's input
    b = 80 * 4 + 512 * 4 + 512 * 4 + 512 * P                         # conv_pre (fp32 in / out), its output's planes
    for s in range(4):
        e = E[s]
        if s == 3:
            b += Ein[s] * P + e * 4 + sum(3 * 8 * e + (4 * e if j else 0) for j in range(3))
            continue
        b += Ein[s] * P + e * P + (0 if residual_from_planes else 4 * e)
        for j in range(3):
            if s == 2 and j == 0 and residual_from_planes:            # the k = 3 ResBlock of stage 2: fused pairs (ev_pair64_mx.h), planes in / planes (or the MRF sum) out
                b += 2 * (e * P + e * P) + e * P + (e * R if mrf_partial_planes else 4 * e)
                continue
            for d in range(3):
                res = e * R if residual_from_planes else 4 * e
                b += 2 * e * P + e * P + res                              # conv1 in / out, conv2 xt in + residual
                if d < 2:
                    b += e * P + (0 if residual_from_planes else 4 * e)
                elif mrf_partial_planes and residual_from_planes and s < 3:
                    # stages with >= 64 channels (stage 2 since round 6): the running sum as a partial plane set (hi plane + remainder codes + their scales =
                    # R bytes per element), the third ResBlock writes the next up-conv's full plane set directly
                    b += (e * R if j else 0) + (e * P if j == 2 else e * R)
                else:
                    b += (4 * e if j else 0) + 4 * e + (e * P if j == 2 else 0)     # running MRF sum in / out (or the stage output + its planes)
    return b + E[3] * 4 + 256 * 4                                    # conv_post


# per-mode layer-wise contracts of the modes' own data flows (fp16: SURVEY 8(d); split precision: fp32 tensors; mx: the residual rebuilt
# from planes, the fused k = 3 pairs of stage 2 and the MRF sums of stages 0-2 as partial plane sets; 4.310 MB with round 3's flow)
VOC_BYTES_PER_FRAME_BY_MODE = {"f16": 2.026e6, "x3": 2 * 2.026e6, "mx": mx_contract_bytes_per_frame(True)}
DEC_FLOP_PER_UTT_1024 = 40.265e9                   # mel decoder at T = 1024
PEAK_MFMA_F16 = 2500.0                             # TFLOP/s dense (MI355X_MICROARCH.md)
PEAK_MFMA_FP4 = 10000.0                            # TFLOP/s dense, block-scaled fp4 / fp6 (MI355X_MICROARCH.md)
PEAK_HBM = 8000.0                                  # GB/s
DTYPE_NAME = {"f16": "f16", "x3": "f16x3 (fp16 hi/lo split, 3 MFMAs per product, fp32 activations)",
              "mx": "f16+mxfp4 (fp32-class activations travelling as plane sets; per product one fp16 MFMA on the hi parts + two block-scaled MFMAs for the cross "
                    "terms: fp4 x fp4 in every generator stage, the decoder's conv-FFN and QKV / output projections, fp4 weights x E5M2 activations in the fused "
                    "32-channel k = 3 pairs; split-precision fp16x3 only in the decoder's attention and LayerNorm-side glue and in the token-rate path)"}
# measured DC-free relative L2 of the waveform against the reference's own outputs (tests/test_gpu_parity.py; profiles/r6_c_parity_report.json: the 8 fixtures of
# weight seed 0; profiles/r6_d_parity_weight_draws.json: the second and third draw of the weights, round 6)
PARITY_LEVEL = {"mx": "<= 4.8e-4 on the 8 reference fixtures of weight seed 0 (zero-mean and trained-like gains included); over three draws of the weights the worst case "
                      "is 8.8e-4 (seed 1, trained-like gains: that draw is 1.7x harder in every mode) -- bar 1e-3",
                "fast": "2.4e-3 ... 4.3e-3 on zero-mean audio (bar 1e-3: NOT met)", "strict": "<= 5e-6"}


def reference_cpu_record():
    """The reference ITSELF timed in the build container (tools/time_reference.py -> profiles/r5_reference_cpu.json, kind "reference"): 64 phonemes =
    BASELINE configs[0], 256 phonemes = the utterance length of configs[1].  /root/reference cannot travel to the GPU box, so this is a committed
    measurement quoted beside the live numbers, never `value`."""
    path = os.path.join(ROOT, "profiles", "r5_reference_cpu.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    out = dict(kind=rec.get("kind"), cores=rec.get("cores"), host=rec.get("host", {}).get("cpu"), source="profiles/r5_reference_cpu.json (tools/time_reference.py, build container)")
    for k, v in rec.get("results", {}).items():
        out[k] = dict(frames_per_s=v["frames_per_s"], ms_per_utterance=v["ms_per_utterance_median"], x_realtime=v["x_realtime"])
    return out


def decoder_flops(frames):
    """Algorithmic FLOPs of the 4-layer mel decoder for one utterance of T frames (SURVEY.md section 8(d) formula)."""
    T = float(frames)
    return 2.0 * 4.0 * T * (4 * 384 ** 2 + 2 * T * 384 + 2 * 384 * 1536 * 3)


def csrc_hash():
    """Content hash of the kernel sources: profiles/latest_hbm_traffic.json carries the hash it was measured on (tools/profile_summary.py), and
    a `traffic` figure is only printed for the same sources (the GPU box has no .git, so a commit id cannot be read there)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "emotivoice_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(n_utts, phonemes, threads=16):
    """The CPU oracle (a port of the reference path) timed on this box's host cores: B = 1 per utterance,
    the only batch size the reference's call sites use."""
    import torch
    from oracle import EVShapes, jets_forward
    from oracle.jets_oracle import to_torch_sd
    from emotivoice_amd.synthetic import synth_inputs, synth_state_dict
    # 256 torch threads on the GPU box's 2x64-core EPYC run ~100x slower than 16 (oversubscription); the reference
    # itself pins a handful of threads per worker (inference_tts.py:186 uses 4).  16 is what we time and report.
    cores = min(os.cpu_count() or 1, threads)
    torch.set_num_threads(cores)
    sd = to_torch_sd(synth_state_dict(0, "bench"))
    utts = synth_inputs(1, [phonemes] * (n_utts + 1), None)
    jets_forward(sd, utts[0]["ling"], 0, utts[0]["style"], utts[0]["content"], EVShapes())   # warm-up
    frames = 0
    t0 = time.perf_counter()
    done = 0
    for u in utts[1:]:
        o = jets_forward(sd, u["ling"], 0, u["style"], u["content"], EVShapes())
        frames += int(o["mel_len"])
        done += 1
        if time.perf_counter() - t0 > 25.0:      # bounded sample: ~10-30 s of CPU work
            break
    dt = time.perf_counter() - t0
    return dict(value=frames / dt, unit="mel-frames/s", cores=cores, kind="port",
                sample="%d utterances x %d phonemes, B=1 loop, fp32 torch-CPU oracle, %.1f s" % (done, phonemes, dt),
                # (one figure, one source: the committed record of tools/time_reference.py -- SURVEY section 6's preliminary 628 frames/s is superseded by it)
                reference_measured_in_build_container=dict(note="the reference's own JETSGenerator module (kind 'reference') timed on the BUILD CONTAINER's 8 cores, "
                                                                "not on this box; /root/reference does not exist on the GPU box", **(reference_cpu_record() or {})))


class Workload:
    """Device-resident inputs of one rank + the callable that runs one step."""

    def __init__(self, args, eng, rank, dev, torch, _ffi):
        from emotivoice_amd.synthetic import synth_inputs
        self.eng, self.mode = eng, args.mode
        self.calls = []          # one closure per sub-batch
        self.desc = {}
        if args.mode == "vocoder":
            nb = args.batch or 128
            rng = np.random.default_rng(9 + rank)
            base = (1.25 * rng.standard_normal((8, 80, 1024)) + 0.08).astype(np.float16)       # mel statistics of the oracle AM
            mel = torch.from_numpy(np.ascontiguousarray(np.concatenate([base[i % 8].ravel() for i in range(nb)]))).to(dev)
            lens = np.full(nb, 1024, np.int32)
            self.keep = [mel]
            self.calls.append(lambda: eng.vocoder_raw(nb, mel.data_ptr(), True, lens, _ffi.EV_FLAG_DEVICE_INPUTS))
            self.desc = dict(workload="configs[4]: %d pre-computed 80x1024 fp16 mels per GPU, vocoder only" % nb, batch_per_gpu=nb)
            self.utts_per_step = nb
            return
        if args.mode == "ragged":
            nb = args.batch or 256
            lens = [64 + ((i + rank * nb) * 7919) % 449 for i in range(nb)]
            spk = [(i + rank * nb) % 2000 for i in range(nb)]
            groups = [(lens, spk)]
            self.desc = dict(workload="configs[2]: batch=%d, mixed 64-512 phoneme lengths, 2000-speaker round-robin, AM+vocoder" % nb,
                             batch_per_gpu=nb, phonemes_per_step=int(sum(lens)))
        else:
            nb = args.batch or 32
            nsub = args.sub_batches
            groups = [([args.phonemes] * nb, [0] * nb) for _ in range(nsub)]
            cfgname = "configs[1]" if nsub == 1 else "configs[3] share"
            self.desc = dict(workload="%s: %d x batch=%d x %d-phoneme synthetic utterances per GPU per step, 1 speaker, AM+vocoder end-to-end, "
                                      "4 frames/phoneme" % (cfgname, nsub, nb, args.phonemes),
                             batch_per_gpu=nb * nsub, sub_batches=nsub, phonemes=args.phonemes)
        self.keep = []
        self.utts_per_step = 0
        for gi, (lens, spk) in enumerate(groups):
            utts = synth_inputs(1 + rank * 64 + gi, lens, spk)
            B = len(utts)
            ling = torch.from_numpy(np.concatenate([u["ling"] for u in utts])).to(dev)
            cu = np.zeros(B + 1, np.int32)
            cu[1:] = np.cumsum(lens)
            spk_t = torch.tensor(spk, dtype=torch.int64, device=dev)
            style = torch.from_numpy(np.stack([u["style"] for u in utts])).to(dev)
            content = torch.from_numpy(np.stack([u["content"] for u in utts])).to(dev)
            self.keep += [ling, spk_t, style, content]
            self.calls.append(lambda B=B, ling=ling, cu=cu, spk_t=spk_t, style=style, content=content:
                              eng.synthesize_raw(B, ling.data_ptr(), cu, spk_t.data_ptr(), style.data_ptr(), content.data_ptr(), 1.0,
                                                 _ffi.EV_FLAG_DEVICE_INPUTS))
            self.utts_per_step += B

    def step(self):
        frames = 0
        for c in self.calls:
            frames += int(c().total_frames)
        return frames


def power_sample(work, torch, seconds=2.5):
    """Clock and socket power while the workload loops (rocm-smi polled from a thread; the readings lag the load by a few hundred ms, so the first 0.7 s are
    dropped).  Outside the timed region.  None when rocm-smi is not there.  Round 6 finding (profiles/r6_j_*): every kernel family of the generator runs AT the
    1400 W board limit, at 1.63-2.03 GHz instead of 2.4 -- the roofs quoted against the nominal clock have a ceiling of 0.68-0.85 under this load."""
    import re
    import shutil
    import subprocess
    import threading
    import time
    if not shutil.which("rocm-smi"):
        return None
    samples, stop = [], [False]

    def poll():
        while not stop[0]:
            t = time.perf_counter()
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                return
            m1, m2 = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out), re.search(r"Power \(W\): ([\d.]+)", out)
            if m1 and m2:
                samples.append((t, int(m1.group(1)), float(m2.group(1))))

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        work.step()
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    stop[0] = True
    th.join(timeout=6)
    mine = [x for x in samples if t0 + 0.7 < x[0] < t1 - 0.05]
    if len(mine) < 3:
        return None
    return dict(sclk_MHz=int(np.median([x[1] for x in mine])), socket_W=round(float(np.median([x[2] for x in mine])), 0), samples=len(mine), nominal_sclk_MHz=2400,
                note="rocm-smi polled while the workload loops for %.1f s outside the timed region; board limit 1400 W" % seconds)


def b1_latency(eng, phonemes=(64, 256)):
    """Single-utterance latency (the reference's own call pattern, B = 1; BASELINE configs[0] is 64 phonemes): host-to-host wall time of one
    ev_synthesize with host inputs, best of 20.  Measured BEFORE the throughput loop: right after it the chip still runs at its sustained-load
    clocks (~1.9 instead of 2.4 GHz) and the same call measures ~20 % longer, which says nothing about a serving process answering single requests."""
    from emotivoice_amd.synthetic import synth_inputs
    lat = {}
    for nph in phonemes:
        u = synth_inputs(99, [nph], None)[0]
        ling1 = np.ascontiguousarray(u["ling"]); cu1 = np.array([0, nph], np.int32)
        spk1 = np.zeros(1, np.int64); st1 = np.ascontiguousarray(u["style"]); ct1 = np.ascontiguousarray(u["content"])
        best = 1e9
        for it in range(23):
            t1 = time.perf_counter()
            r1 = eng.synthesize_raw(1, ling1.ctypes.data, cu1, spk1.ctypes.data, st1.ctypes.data, ct1.ctypes.data, 1.0, 0)
            dtl = time.perf_counter() - t1
            if it >= 3:
                best = min(best, dtl)
        lat["b1_%dph_ms" % nph] = round(best * 1e3, 3)
        lat["b1_%dph_x_realtime" % nph] = round(int(r1.total_frames) * 256 / 16000 / best, 1)
    # BASELINE configs[0] names the CPU PyTorch reference path at 64 phonemes: the reference's own module, timed in the build container, beside the GPU call
    ref = reference_cpu_record()
    if ref:
        lat["cpu_reference"] = ref
        for nph in phonemes:
            r = ref.get("%dph" % nph)
            if r:
                # (a committed timing from ANOTHER machine -- the build container's 8 cores -- over this box's live GPU latency: named for what it is)
                lat["b1_%dph_speedup_vs_build_container_reference" % nph] = round(r["ms_per_utterance"] / lat["b1_%dph_ms" % nph], 1)
    return lat


def timed(work, steps, warmup, sync):
    for _ in range(warmup):
        work.step()
    sync()
    t0 = time.perf_counter()
    frames = 0
    for _ in range(steps):
        frames += work.step()
    sync()
    return frames, time.perf_counter() - t0


# kernel families of the generator per precision mode: (stat name, MFMA-equivalents issued per algorithmic product, description)
VOC_FAMILIES = {
    "voc_conv_gemm_f16": (1.0, "conv_gemm_kernel<f16> + conv_gemm_phased_kernel: one fp16 MFMA per product"),
    "voc_resblock_pair_c32": (1.0, "resblock_pair_c32_kernel (fused pair, fp16)"),
    "voc_resblock_pair_c64": (1.0, "resblock_pair_c64_kernel (fused pair, fp16)"),
    "voc_conv_gemm_x3": (3.0, "conv_gemm_x3_kernel / conv_gemm_split_kernel: three fp16 MFMAs per product"),
    "voc_conv_gemm_mx": (1.5, "conv_gemm_mx_kernel: one fp16 MFMA + two block-scaled fp4 MFMAs (4x rate) per product, operand planes from the producer's epilogue"),
    "voc_conv_c64_mx": (1.5, "conv_c64_mx_kernel (stage 2, C = 64: two taps per fp4 MFMA, plane sets in / out)"),
    "voc_resblock_pair_c64_mx": (1.5, "resblock_pair_c64_mx_kernel (fused k = 3 pair of stage 2, plane sets in / out, xt and the residual never cross HBM)"),
    "voc_resblock_pair_c32_mx": (1.5, "resblock_pair_c32_e5_kernel at k = 3 (fp4 weights x E5M2 activations) / resblock_pair_c32_mx2_kernel at k = 7, 11 (fused pair, fp16 + block-scaled MFMAs, fp32 in / out)"),
}
DOMINANT = {"f16": "voc_conv_gemm_f16", "x3": "voc_conv_gemm_x3", "mx": "voc_conv_gemm_mx"}


def roofline_block(eng, work, torch, dump=None):
    """One extra profiled step (hipEvents on the engine's stream around every launch, single stream) -> per-family accounting.
    `achieved` of the dominant family = the ALGORITHMIC FLOPs of exactly its launches (one product per multiply-add, transposed convs at
    their two real taps) / the sum of their hipEvent durations; `issued_equiv_*` scales that by the MFMA-equivalents the family's
    arithmetic issues per product (3 for the split precision, 1.5 for MX: the fp4 MFMA runs at 4x the fp16 rate)."""
    eng.set_profiling(True)
    res = work.calls[0]()
    torch.cuda.synchronize()
    stats = {s["name"]: s for s in eng.kernel_stats()}
    stages = eng.timings()
    mode = eng.vocoder_precision
    if dump:                                               # --dump-launches: per-launch table (kernel, shape, ms, TF/s, GB/s) for profiles/
        recs = eng.launch_records()
        for r in recs:
            r["TFLOPs"] = round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1) if r["ms"] > 0 and r["flops"] > 0 else None
            r["GBps"] = round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1) if r["ms"] > 0 and r["bytes"] > 0 else None
        with open(dump + ("" if mode == "f16" else "." + mode), "w") as f:
            json.dump(recs, f, indent=0)
    eng.set_profiling(False)
    f1 = int(res.total_frames)
    B = int(res.batch)
    kernels = {k: dict(ms=round(v["ms"], 3), launches=v["launches"]) for k, v in stats.items()}
    dom = stats.get(DOMINANT[mode])
    roof = None
    if dom and dom["ms"] > 0:
        units, desc = VOC_FAMILIES[DOMINANT[mode]]
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        fams = {}
        voc_ms = voc_issued = 0.0
        for name, (u, d) in VOC_FAMILIES.items():
            st = stats.get(name)
            if not st or st["ms"] <= 0:
                continue
            tf = st["flops"] / (st["ms"] * 1e-3) / 1e12
            fams[name] = dict(ms=round(st["ms"], 3), launches=st["launches"], algorithmic_TFLOPs=round(tf, 2), mfma_units_per_product=u,
                              issued_equiv_frac=round(u * tf / PEAK_MFMA_F16, 4))
            voc_ms += st["ms"]
            voc_issued += u * st["flops"]
        voc_pmc = None               # PMC-measured HBM bytes of the whole generator per mel frame, this mode
        traffic = tcommit = None    # HBM bytes per launch of the dominant family from the PMC passes (separate rocprofv3 --pmc runs, see profiles/)
        tnote = "no PMC passes for this workload / mode"
        tpath = os.path.join(ROOT, "profiles", "latest_hbm_traffic.json")
        if os.path.exists(tpath) and work.mode == "am_vocoder" and B == 32:
            tall = json.load(open(tpath))
            t = tall.get(mode, tall if mode == "f16" and "hbm_bytes_per_launch" in tall else {})          # one entry per precision mode (tools/profile_summary.py)
            tcommit = tall.get("csrc_hash")
            if tcommit and tcommit == csrc_hash():
                traffic = t.get("hbm_bytes_per_launch")
                voc_pmc = t.get("vocoder_hbm_bytes_per_frame")
                tnote = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches from separate rocprofv3 --pmc passes of these kernel sources "
                         "(profiles/latest_hbm_traffic.json, csrc_hash matches)")
            else:
                tnote = ("profiles/latest_hbm_traffic.json was measured on other kernel sources (csrc_hash %s, now %s): stale, not reported -- "
                         "re-run tools/run_profiles.sh" % (tcommit, csrc_hash()))
        # dense MFMA peak of the family's ARITHMETIC per algorithmic product: fp16 2.5 PF/s; split precision = three fp16 MFMAs -> 2.5 / 3;
        # mx = one fp16 MFMA (2.5 PF/s) + two fp4 MFMAs (10 PF/s, MI355X_MICROARCH.md) -> 1 / (1 / 2.5 + 2 / 10) = 1.667 PF/s
        peak = {1.0: PEAK_MFMA_F16, 3.0: PEAK_MFMA_F16 / 3.0, 1.5: 1.0 / (1.0 / PEAK_MFMA_F16 + 2.0 / PEAK_MFMA_FP4)}[units]
        roof = dict(bound="mfma", kernel=desc + " (HiFi-GAN Conv1d / ConvTranspose1d launches of this family)",
                    achieved=round(achieved, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(achieved / peak, 4),
                    peak_note="dense MFMA peak of this arithmetic per algorithmic product (fp16: 2500; 3 x fp16: 833; fp16 + 2 x fp4: 1667 TFLOP/s)",
                    frac_of_fp16_peak=round(achieved / PEAK_MFMA_F16, 4),
                    traffic=traffic, traffic_note=tnote, traffic_commit=tcommit, launches=dom["launches"], avg_launch_ms=round(dom["ms"] / dom["launches"], 4),
                    algorithmic_flop_per_launch=round(dom["flops"] / dom["launches"], 1),
                    mfma_units_per_product=units, issued_equiv_TFLOPs=round(units * achieved, 2), issued_equiv_frac=round(units * achieved / PEAK_MFMA_F16, 4),
                    note="achieved / frac count ALGORITHMIC FLOPs; issued_equiv_* = what the matrix pipes execute in fp16-MFMA equivalents",
                    families=fams,
                    all_vocoder_convs=dict(ms=round(voc_ms, 3), mfma_TFLOPs=round(VOC_CONV_FLOP_PER_FRAME * f1 / (voc_ms * 1e-3) / 1e12, 2),
                                           issued_equiv_frac=round(voc_issued / (voc_ms * 1e-3) / 1e12 / PEAK_MFMA_F16, 4),
                                           hbm_algorithmic_GBps=round(VOC_BYTES_PER_FRAME * f1 / (voc_ms * 1e-3) / 1e9, 1),
                                           hbm_algorithmic_frac=round(VOC_BYTES_PER_FRAME * f1 / (voc_ms * 1e-3) / 1e9 / PEAK_HBM, 4),
                                           hbm_algorithmic_note="SURVEY.md section 8(d)'s numerator in EVERY mode: 2.026 MB per mel frame (layer-wise fp16 contract) / "
                                                                "time of the generator's conv launches / 8 TB/s -- the figure to hold against north_star's >= 0.40; the "
                                                                "hbm_contract_* and hbm_pmc_* fields below divide bytes this mode moves, not useful work",
                                           hbm_contract_GBps=round(VOC_BYTES_PER_FRAME_BY_MODE[mode] * f1 / (voc_ms * 1e-3) / 1e9, 1),
                                           hbm_contract_frac=round(VOC_BYTES_PER_FRAME_BY_MODE[mode] * f1 / (voc_ms * 1e-3) / 1e9 / PEAK_HBM, 4),
                                           hbm_contract_bytes_per_frame=VOC_BYTES_PER_FRAME_BY_MODE[mode],
                                           hbm_contract_note="this mode's layer-wise byte contract (fp16: 2.026 MB / frame; split precision: fp32 tensors, 2x; "
                                                             "mx: plane sets, residuals rebuilt from them, running MRF sums as partial plane sets in stages 0-2 and fp32 in stage 3, fused k = 3 pairs at C = 64: 3.16 MB / frame)"))
        if voc_pmc:
            roof["all_vocoder_convs"].update(hbm_pmc_bytes_per_frame=round(voc_pmc, 1), hbm_pmc_GBps=round(voc_pmc * f1 / (voc_ms * 1e-3) / 1e9, 1),
                                             hbm_pmc_frac=round(voc_pmc * f1 / (voc_ms * 1e-3) / 1e9 / PEAK_HBM, 4),
                                             hbm_pmc_note="bytes the generator actually moves (rocprofv3 FETCH_SIZE / WRITE_SIZE passes, calibrated) / time")
    if roof is not None and "decoder" in stages and stages["decoder"] > 0:
        dflop = decoder_flops(f1 / max(B, 1)) * B if work.mode == "am_vocoder" else None
        if dflop:
            tf = dflop / (stages["decoder"] * 1e-3) / 1e12
            roof["mel_decoder"] = dict(ms=round(stages["decoder"], 3), mfma_TFLOPs=round(tf, 2), mfma_frac=round(tf / PEAK_MFMA_F16, 4),
                                       precision=eng.decoder_precision)
    return roof, stages, kernels


def bench_style(args, torch):
    """--mode style: the SimBERT prompt / content encoder on the device (the reference runs it on the CPU, two calls
    per utterance): texts of 64 tokens, pooled outputs per second, with the CPU oracle (the same BERT forward in torch fp32) beside it."""
    from emotivoice_amd.engine import EVEngine
    from emotivoice_amd.packer import pack_bert_state_dict
    from emotivoice_amd.synthetic import synth_bert_state_dict, synth_token_ids
    sd = synth_bert_state_dict(0)
    blob, _, cfg = pack_bert_state_dict(sd)
    eng = EVEngine(device_id=0)
    eng.style_load(blob, cfg)
    nb = args.batch or 64
    ids = synth_token_ids(5, [64] * nb)
    for _ in range(args.warmup):
        eng.style_embed(ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = eng.style_embed(ids)
    dt = time.perf_counter() - t0
    one = [ids[0]]
    best = 1e9
    for _ in range(10):
        t1 = time.perf_counter(); eng.style_embed(one); best = min(best, time.perf_counter() - t1)
    line = {"metric": "style_embeddings_per_sec", "value": round(nb * args.steps / dt, 1), "unit": "texts/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3 (split precision, fp32-class)", "data": "synthetic",
            "config": {"workload": "SimBERT (BERT-base 12 x 768) pooled_output of %d texts x 64 tokens per step, host ids in / host embeddings out" % nb},
            "latency": {"one_text_64_tokens_ms": round(best * 1e3, 3)}, "roofline": None}
    if args.cpu_utts > 0:
        from oracle.bert_oracle import bert_pooled_output
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        bert_pooled_output(sd, ids[0])
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 10.0 and n < 64:
            bert_pooled_output(sd, ids[n % nb]); n += 1
        cdt = time.perf_counter() - t0
        line["cpu_baseline"] = dict(value=n / cdt, unit="texts/s", cores=min(os.cpu_count() or 1, 16), kind="port",
                                    sample="%d texts x 64 tokens, one per call (the reference's call pattern), torch fp32, %.1f s" % (n, cdt))
        ref = bert_pooled_output(sd, ids[0]).numpy()
        line["parity_rel_l2_vs_oracle"] = float(np.linalg.norm(out[0] - ref) / np.linalg.norm(ref))
    print(json.dumps(line))


def bench_pipeline(args, torch):
    """--mode pipeline (not a BASELINE config): the HOST pipeline end to end at B = 32 -- text line -> G2P -> phoneme ids,
    prompt + content -> WordPiece ids -> SimBERT on the device, ev_synthesize with host inputs (mx precision), int16 waveform back on the
    host -- one process, one GPU, so that the limiter of the 8-GPU target (host feeding) is a measured number.
    Stand-ins for what is not in the image: tools/g2p_standin.py (the reference's English lexicon path at realistic cost), a WordPiece
    vocabulary built from the same lexicon, seeded synthetic SimBERT / generator weights."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from g2p_standin import make_g2p, make_lexicon, make_text, token_table
    from emotivoice_amd import _ffi
    from emotivoice_amd.engine import EVEngine
    from emotivoice_amd.frontend_pool import FrontendPool
    from emotivoice_amd.packer import pack_bert_state_dict, pack_state_dict
    from emotivoice_amd.synthetic import synth_bert_state_dict, synth_state_dict
    from emotivoice_amd.wordpiece import WordPieceTokenizer
    nb = args.batch or 32
    lex = make_lexicon()
    g2p = make_g2p(lex)
    tok2id = token_table()
    texts = make_text(lex, nb * (args.steps + args.warmup), words_per_line=14)
    prompts = ["happy", "sad", "angry", "excited"]
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", ",", "."] + list(lex.keys())[:13000] + prompts
        f.write("\n".join(vocab) + "\n")
        vpath = f.name
    tok = WordPieceTokenizer(vpath)
    pool = FrontendPool(g2p, workers=args.frontend_workers) if args.frontend_workers > 1 else None      # forks BEFORE the HIP runtime starts
    eng = EVEngine(device_id=0, precision=args.precision)
    eng.load_blob(*pack_state_dict(synth_state_dict(0, "parity")))
    bblob, _, bcfg = pack_bert_state_dict(synth_bert_state_dict(0))
    eng.style_load(bblob, bcfg)
    parts = dict(g2p=0.0, ids=0.0, tokenize=0.0, simbert=0.0, synth=0.0, d2h=0.0)
    frames = utts = 0
    t_all = None
    for step in range(args.warmup + args.steps):
        if step == args.warmup:
            torch.cuda.synchronize()
            parts = {k: 0.0 for k in parts}
            frames = utts = 0
            t_all = time.perf_counter()
        lines = texts[step * nb:(step + 1) * nb]
        t0 = time.perf_counter()
        phon = pool.map(lines) if pool is not None else [g2p(t) for t in lines]
        t1 = time.perf_counter()
        ling = [np.array([tok2id[x] % 502 for x in ph.split()], np.int64) for ph in phon]
        t2 = time.perf_counter()
        ids = [np.asarray(tok.encode(t, max_length=512), np.int64) for t in [prompts[i % 4] for i in range(nb)] + lines]
        t3 = time.perf_counter()
        emb = eng.style_embed(ids)                                   # (2 nb, 768): prompt rows, then content rows
        t4 = time.perf_counter()
        cu = np.zeros(nb + 1, np.int32); cu[1:] = np.cumsum([len(x) for x in ling])
        flat = np.ascontiguousarray(np.concatenate(ling)); spk = np.arange(nb, dtype=np.int64) % 2014
        style, content = np.ascontiguousarray(emb[:nb]), np.ascontiguousarray(emb[nb:])
        res = eng.synthesize_raw(nb, flat.ctypes.data, cu, spk.ctypes.data, style.ctypes.data, content.ctypes.data, 1.0, _ffi.EV_FLAG_WANT_INT16)
        t5 = time.perf_counter()
        wav = eng.d2h(res.wav_i16, (res.total_samples,), np.int16)
        t6 = time.perf_counter()
        for k, v in zip(parts, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            parts[k] += v
        frames += int(res.total_frames); utts += nb
    dt = time.perf_counter() - t_all
    if pool is not None:
        pool.close()
    os.unlink(vpath)
    assert wav.size == int(res.total_samples) and np.abs(wav).max() > 0
    print(json.dumps({
        "metric": "host_pipeline_utterances_per_sec", "value": round(utts / dt, 1), "unit": "utterances/s", "mel_frames_per_sec": round(frames / dt, 1),
        "x_realtime": round(frames / dt * 256 / 16000, 1), "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "data": "synthetic", "precision": args.precision,
        "config": {"workload": "text line -> int16 wav on the host, B = %d lines of 14 words (~%d phonemes) per step, one process" % (nb, int(np.mean([len(x) for x in ling]))),
                   "frontend_workers": args.frontend_workers},
        "host_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in parts.items()},
        "note": "g2p / ids / tokenize are pure host time; simbert, synth and d2h include their device time (the calls are synchronous)"}))


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_command(n, argv):
    """The launcher command `python bench.py --gpus N ...` turns into when no launcher started it: torch.distributed.run with
    one rank per GPU on this node, rendezvous on 127.0.0.1 (the reference's analogue is the process-per-GPU fan-out of
    inference_tts.py:178-220)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def rendezvous_only(rank, world):
    """Launch check without a GPU: every rank joins the group, rank 0 prints what the real run would print about the job's shape."""
    import torch
    ranks_seen = [0]
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        rk = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(rk, torch.tensor([rank], dtype=torch.int64))
        ranks_seen = [int(t.item()) for t in rk]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "ranks_seen": ranks_seen}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="am_vocoder", choices=["am_vocoder", "ragged", "vocoder", "style", "pipeline"])
    ap.add_argument("--frontend-workers", type=int, default=1, help="--mode pipeline: G2P worker processes (1 = in the synthesis process, like the reference)")
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU per (sub-)batch (default: 32 / 256 / 128 by mode)")
    ap.add_argument("--sub-batches", type=int, default=0, help="am_vocoder: sub-batches per step (default 1 at N = 1, 8 at N > 1 = configs[3])")
    ap.add_argument("--phonemes", type=int, default=256)
    ap.add_argument("--cpu-utts", type=int, default=12, help="utterances for the CPU baseline sample (0 = skip); 12 x 256 phonemes = 12-15 s of host work on 16 threads")
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch threads of the CPU baseline (more are slower on the 2 x 64-core box)")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch table of the profiled step to this path (+ .<mode> for the other precisions)")
    ap.add_argument("--chunk-mb", type=int, default=0, help="ev_config.vocoder_chunk_mb (tuning; 0 = whole tensors)")
    ap.add_argument("--voc-streams", type=int, default=0, help="ev_config.vocoder_streams (tuning; 0 = engine default)")
    ap.add_argument("--mx-residual", default="planes", choices=["planes", "fp32"], help="ev_config.mx_residual (A/B: round 3's fp32 residual stream)")
    ap.add_argument("--mx-mrf", default="planes", choices=["planes", "fp32"], help="ev_config.mx_mrf (A/B: fp32 running MRF sum)")
    ap.add_argument("--decoder-ln", default="planes", choices=["planes", "fp32"], help="ev_config.decoder_ln_planes (A/B: fp32 LayerNorm output + planes pass)")
    ap.add_argument("--mx-act-format", default="e5m2", choices=["e5m2", "fp4"], help="ev_config.mx_act_format (A/B: activation operand of the cross terms in the fused 32-channel k = 3 pairs)")
    ap.add_argument("--token-splitk", default="on", choices=["on", "off"], help="ev_config.token_splitk (A/B: the token-rate long-K GEMMs in one pass)")
    ap.add_argument("--precision", default="mx", choices=["mx", "fast", "strict"],
                    help="frame-rate path: mx (default, the contract mode: waveform <= 1e-3 on every fixture) = fp32 activations, one fp16 MFMA + "
                         "two block-scaled fp4 MFMAs per product; fast = fp16 MFMA operands / fp16 activations (2.4e-3 on zero-mean audio); "
                         "strict = split precision (3 fp16 MFMAs per product, ~1e-6)")
    ap.add_argument("--decoder-precision", default=None, choices=["f16", "f32", "x3"])
    ap.add_argument("--no-other-precision", action="store_true", help="skip the extra timed passes in the other precisions (N = 1)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests)")
    ap.add_argument("--force-device", type=int, default=-1, help="debug: put every rank on this device (with --backend gloo)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch check: the ranks meet (gloo), rank 0 prints n_gpus / ranks_seen, nothing touches a GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher of N ranks (one per GPU) running this same command line
        raise SystemExit(subprocess.call(spawn_command(args.gpus, sys.argv[1:])))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.rendezvous_only:
        return rendezvous_only(rank, world)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: emotivoice_amd has no CPU fallback")
    if args.force_device < 0 and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d device(s) visible" % (world, torch.cuda.device_count()))
    if args.force_device >= 0:
        local_rank = args.force_device
    if args.sub_batches <= 0:
        args.sub_batches = 1 if world == 1 else 8
    torch.cuda.set_device(local_rank)
    if args.mode == "style":
        return bench_style(args, torch)
    if args.mode == "pipeline":
        return bench_pipeline(args, torch)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from emotivoice_amd import _ffi
    from emotivoice_amd.engine import EVEngine
    from emotivoice_amd.sharding import broadcast_blob

    # weights: rank 0 packs, everyone receives the blob with one broadcast (RCCL over xGMI), borrowed in place
    dur_mode = "parity" if args.mode == "ragged" else "bench"
    blob_t = broadcast_blob(rank, world, local_rank, dist, dur_mode=dur_mode)
    dev = torch.device("cuda", local_rank)

    def make_engine(precision, decoder_precision=None):
        e = EVEngine(device_id=local_rank, precision=precision, decoder_precision=decoder_precision,
                     vocoder_chunk_mb=args.chunk_mb, vocoder_streams=args.voc_streams, mx_residual=args.mx_residual,
                     mx_mrf=args.mx_mrf, decoder_ln=args.decoder_ln, token_splitk=args.token_splitk == "on", mx_act_format=args.mx_act_format)
        e.load_blob_device(blob_t.data_ptr(), blob_t.numel(), keepalive=blob_t)
        return e

    eng = make_engine(args.precision, args.decoder_precision)
    lat0 = b1_latency(eng) if rank == 0 and args.mode == "am_vocoder" else {}
    work = Workload(args, eng, rank, dev, torch, _ffi)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    frames, dt = timed(work, args.steps, args.warmup, sync)
    ranks_seen = [0]
    per_rank = [dict(rank=0, frames=int(frames), s=round(dt, 6))]
    if dist is not None:
        # every rank's own (rank, frames, seconds): a straggler is visible in the line; value = all frames / the slowest rank's time
        mine = torch.tensor([float(rank), float(frames), dt], device=dev, dtype=torch.float64)
        allr = [torch.zeros(3, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [dict(rank=int(t[0].item()), frames=int(t[1].item()), s=round(float(t[2].item()), 6)) for t in allr]
        ranks_seen = [r["rank"] for r in per_rank]
        frames = sum(r["frames"] for r in per_rank)
        dt = max(r["s"] for r in per_rank)

    roof, stages, kernels, lat, other, power = None, {}, {}, lat0, None, None
    if rank == 0:
        roof, stages, kernels = roofline_block(eng, work, torch, args.dump_launches)
        # (behind the profiled step: 2.5 s of sustained load in front of it cost its per-launch numbers 5-10 % -- the chip throttles further as it warms)
        power = power_sample(work, torch) if world == 1 else None
        # the same workload in the other precisions (one engine at a time: the previous one's workspace is released first)
        if world == 1 and not args.no_other_precision and args.decoder_precision is None:
            eng.close()
            other = []
            for op in [m for m in ("mx", "fast", "strict") if m != args.precision]:
                eng2 = make_engine(op)
                work2 = Workload(args, eng2, rank, dev, torch, _ffi)
                n2 = max(3, args.steps // 2)
                f2, dt2 = timed(work2, n2, 2, lambda: torch.cuda.synchronize())
                roof2, stages2, _ = roofline_block(eng2, work2, torch, args.dump_launches)
                other.append(dict(precision=op, waveform_rel_l2_vs_reference=PARITY_LEVEL[op], value=round(f2 / dt2, 1), unit="mel-frames/s",
                                  x_realtime=round(f2 / dt2 * 256 / 16000, 1), ms_per_step=round(dt2 / n2 * 1e3, 3), roofline=roof2,
                                  stage_ms={k: round(v, 3) for k, v in stages2.items()}))
                eng2.close()
    if rank != 0:
        # stay in the group until rank 0 has printed the line (it profiles one more step first): a launcher that tears the job down when the first
        # rank exits must not lose the result, and a rank that died early shows up as a hang / error instead of a short line
        dist.barrier()
        dist.destroy_process_group()
        return

    sr, hop = 16000, 256
    value = frames / dt
    cfg = dict(work.desc)
    cfg.update(global_batch=work.utts_per_step * world, parallelism="utterance-sharded x%d" % world, precision=args.precision,
               token_rate_precision="f32 (split fp16x3 GEMMs)", frame_rate_precision="%s/%s" % (eng.decoder_precision, eng.vocoder_precision))
    line = {
        "metric": "mel_frames_per_sec", "value": round(value, 1), "unit": "mel-frames/s",
        "x_realtime": round(value * hop / sr, 1),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_NAME[eng.vocoder_precision], "precision": args.precision, "waveform_rel_l2_vs_reference": PARITY_LEVEL[args.precision],
        "data": "synthetic", "ranks_seen": ranks_seen, "ranks": per_rank, "config": cfg,
        "latency": lat, "roofline": roof, "stage_ms": {k: round(v, 3) for k, v in stages.items()}, "kernels_ms": kernels,
        "other_precision": other,
    }
    if power and roof:
        clk = power["sclk_MHz"] / 2400.0
        line["power"] = power
        roof["frac_at_measured_clock"] = round(roof["frac"] / clk, 4)
        roof["measured_clock_note"] = ("the forward runs power-limited: sclk %d MHz at %d W under load (line.power); frac_at_measured_clock = frac x 2400 / sclk -- "
                                       "what the dominant family reaches of its arithmetic's roof at the clock the chip actually sustains" % (power["sclk_MHz"], int(power["socket_W"])))
    if world == 1 and args.cpu_utts > 0:
        line["cpu_baseline"] = cpu_baseline(args.cpu_utts, args.phonemes, args.cpu_threads)
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
