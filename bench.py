#!/usr/bin/env python3
"""Throughput of the EmotiVoice hot path (JETSGenerator.forward = acoustic model + HiFi-GAN) on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload = BASELINE.json configs[1]: batch 32 x 256 synthetic phonemes, one speaker, AM + vocoder end to
end, seeded synthetic weights whose duration head gives exactly 4 frames / phoneme (1024 frames = 16.384 s
of audio per utterance).  One "step" = one ev_synthesize call over the rank's batch with the inputs already
resident in HBM.  N > 1: utterances are sharded (weak scaling, per-GPU batch fixed); the only collective is
the start-up broadcast of the packed weight blob over RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per mel frame (SURVEY.md section 8(d) / BASELINE.md section 3)
VOC_CONV_FLOP_PER_FRAME = 614.105e6 - 0.115e6      # every Conv1d/ConvTranspose1d of the generator except conv_post
VOC_STAGE3_RB_FLOP_PER_FRAME = (9.437 + 22.020 + 34.603) * 1e6   # stage-3 ResBlocks (C = 32): run by the fused pair kernel
VOC_STAGE2_K3_RB_FLOP_PER_FRAME = 18.874e6         # stage-2 k = 3 ResBlock (C = 64): run by the C = 64 fused pair kernel
VOC_BYTES_PER_FRAME = 2.026e6                      # layer-wise fp16 contract
DEC_FLOP_PER_UTT_1024 = 40.265e9                   # mel decoder at T = 1024
AM_FLOP_PER_UTT = 51.43e9
PEAK_MFMA_F16 = 2500.0                             # TFLOP/s dense (MI355X_MICROARCH.md)
PEAK_HBM = 8000.0                                  # GB/s


def cpu_baseline(n_utts, phonemes):
    """The CPU oracle (a port of the reference path) timed on this box's host cores: B = 1 per utterance,
    the only batch size the reference's call sites use."""
    import torch
    from oracle import EVShapes, jets_forward
    from oracle.jets_oracle import to_torch_sd
    from emotivoice_amd.synthetic import synth_inputs, synth_state_dict
    # 256 torch threads on the GPU box's 2x64-core EPYC run ~100x slower than 16 (oversubscription); the reference
    # itself pins a handful of threads per worker (inference_tts.py:186 uses 4).  16 is what we time and report.
    cores = min(os.cpu_count() or 1, int(os.environ.get("EV_CPU_THREADS", "16")))
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
    n_utts = done
    dt = time.perf_counter() - t0
    return dict(value=frames / dt, unit="mel-frames/s", cores=cores, kind="port",
                sample="%d utterances x %d phonemes, B=1 loop, fp32 torch-CPU oracle, %.1f s" % (n_utts, phonemes, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--phonemes", type=int, default=256)
    ap.add_argument("--cpu-utts", type=int, default=8, help="utterances for the CPU baseline sample (0 = skip)")
    ap.add_argument("--precision", default="fast", choices=["fast", "strict"],
                    help="frame-rate path: fast = fp16 MFMA operands (BASELINE configs[1]/[4]: bf16 / fp16); strict = split precision "
                         "(3 fp16 MFMAs per product, fp32 activations)")
    ap.add_argument("--decoder-precision", default=None, choices=["f16", "f32", "x3"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests)")
    ap.add_argument("--force-device", type=int, default=-1, help="debug: put every rank on this device (with --backend gloo)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: emotivoice_amd has no CPU fallback")
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
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
    from emotivoice_amd.synthetic import synth_inputs

    # weights: rank 0 packs, everyone receives the blob with one broadcast (RCCL over xGMI), borrowed in place
    blob_t = broadcast_blob(rank, world, local_rank, dist, dur_mode="bench")
    eng = EVEngine(device_id=local_rank, precision=args.precision, decoder_precision=args.decoder_precision,
                   vocoder_chunk_mb=int(os.environ.get("EV_CHUNK_MB", "0")),       # tuning overrides; 0 = engine default
                   vocoder_streams=int(os.environ.get("EV_VOC_STREAMS", "0")))
    eng.load_blob_device(blob_t.data_ptr(), blob_t.numel(), keepalive=blob_t)

    B, N = args.batch, args.phonemes
    utts = synth_inputs(1 + rank, [N] * B, None)
    dev = torch.device("cuda", local_rank)
    ling = torch.from_numpy(np.concatenate([u["ling"] for u in utts])).to(dev)
    cu = np.arange(B + 1, dtype=np.int32) * N
    spk = torch.zeros(B, dtype=torch.int64, device=dev)
    style = torch.from_numpy(np.stack([u["style"] for u in utts])).to(dev)
    content = torch.from_numpy(np.stack([u["content"] for u in utts])).to(dev)

    def step():
        return eng.synthesize_raw(B, ling.data_ptr(), cu, spk.data_ptr(), style.data_ptr(), content.data_ptr(), 1.0,
                                  _ffi.EV_FLAG_DEVICE_INPUTS)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    sync()
    t0 = time.perf_counter()
    frames = 0
    for _ in range(args.steps):
        res = step()
        frames += int(res.total_frames)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ft = torch.tensor([frames], device=dev, dtype=torch.int64)
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        frames = int(ft.item())

    # one extra profiled step (hipEvents on the engine's stream around every launch) for the roofline block
    roof, stages = None, {}
    if rank == 0:
        eng.set_profiling(True)
        res = step()
        torch.cuda.synchronize()
        stats = {s["name"]: s for s in eng.kernel_stats()}
        stages = eng.timings()
        eng.set_profiling(False)
        f1 = int(res.total_frames)
        voc = stats.get("voc_conv_gemm_f16") or stats.get("voc_conv_gemm_x3")
        traffic = None      # HBM bytes per launch from the PMC passes (separate rocprofv3 --pmc runs, see profiles/)
        tpath = os.path.join(ROOT, "profiles", "latest_hbm_traffic.json")
        if os.path.exists(tpath) and B == 32 and N == 256:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        pair = stats.get("voc_resblock_pair_c32")
        pair64 = stats.get("voc_resblock_pair_c64")
        if voc and voc["ms"] > 0:
            # algorithmic FLOPs of what THIS kernel family executed: all generator convs minus the ResBlocks that ran in the
            # fused pair kernels (stage 3, and the k = 3 ResBlock of stage 2)
            gemm_flop = VOC_CONV_FLOP_PER_FRAME - (VOC_STAGE3_RB_FLOP_PER_FRAME if pair else 0.0) - \
                (VOC_STAGE2_K3_RB_FLOP_PER_FRAME if pair64 else 0.0)
            achieved = gemm_flop * f1 / (voc["ms"] * 1e-3) / 1e12
            voc_ms = voc["ms"] + (pair["ms"] if pair else 0.0) + (pair64["ms"] if pair64 else 0.0)
            roof = dict(bound="mfma", kernel="conv_gemm_kernel<f16> (HiFi-GAN Conv1d/ConvTranspose1d: conv_pre, ups, ResBlocks of stages 0-2 except the fused k=3 one)",
                        achieved=round(achieved, 2), peak=PEAK_MFMA_F16, unit="TFLOP/s", frac=round(achieved / PEAK_MFMA_F16, 4),
                        traffic=traffic,
                        traffic_note="HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches from separate rocprofv3 --pmc passes, profiles/latest_hbm_traffic.json",
                        launches=voc["launches"], avg_launch_ms=round(voc["ms"] / voc["launches"], 4),
                        algorithmic_flop_per_launch=round(gemm_flop * f1 / voc["launches"], 1),
                        all_vocoder_convs=dict(ms=round(voc_ms, 3), mfma_TFLOPs=round(VOC_CONV_FLOP_PER_FRAME * f1 / (voc_ms * 1e-3) / 1e12, 2),
                                               hbm_contract_GBps=round(VOC_BYTES_PER_FRAME * f1 / (voc_ms * 1e-3) / 1e9, 1),
                                               hbm_contract_frac=round(VOC_BYTES_PER_FRAME * f1 / (voc_ms * 1e-3) / 1e9 / PEAK_HBM, 4)))
            if pair:
                roof["fused_pair_kernel"] = dict(ms=round(pair["ms"], 3), launches=pair["launches"],
                                                 mfma_TFLOPs=round(VOC_STAGE3_RB_FLOP_PER_FRAME * f1 / (pair["ms"] * 1e-3) / 1e12, 2))
            if pair64:
                roof["fused_pair_kernel_c64"] = dict(ms=round(pair64["ms"], 3), launches=pair64["launches"],
                                                     mfma_TFLOPs=round(VOC_STAGE2_K3_RB_FLOP_PER_FRAME * f1 / (pair64["ms"] * 1e-3) / 1e12, 2))
        dec = stats.get("dec_f16_gemm") or stats.get("dec_f32_gemm")
        if roof is not None and dec and "decoder" in stages:
            roof["mel_decoder"] = dict(ms=round(stages["decoder"], 3),
                                       mfma_TFLOPs=round(DEC_FLOP_PER_UTT_1024 * B / (stages["decoder"] * 1e-3) / 1e12, 2),
                                       mfma_frac=round(DEC_FLOP_PER_UTT_1024 * B / (stages["decoder"] * 1e-3) / 1e12 / PEAK_MFMA_F16, 4))
        kernels = {k: dict(ms=round(v["ms"], 3), launches=v["launches"]) for k, v in stats.items()}
    # single-utterance latency (the reference's own call pattern, B = 1; BASELINE configs[0] is 64 phonemes): host-to-host
    # wall time of one ev_synthesize with host inputs, best of 20
    lat = {}
    if rank == 0:
        for nph in (64, 256):
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
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    sr, hop = 16000, 256
    value = frames / dt
    line = {
        "metric": "mel_frames_per_sec", "value": round(value, 1), "unit": "mel-frames/s",
        "x_realtime": round(value * hop / sr, 1),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if (eng.decoder_precision, eng.vocoder_precision) == ("f16", "f16") else
                 "decoder %s / vocoder %s (x3 = fp16 hi/lo split, 3 MFMAs per product, fp32 activations)" % (eng.decoder_precision, eng.vocoder_precision),
        "data": "synthetic",
        "config": {"workload": "configs[1]: batch=%d x %d-phoneme synthetic utterances per GPU, 1 speaker, AM+vocoder end-to-end, "
                               "4 frames/phoneme" % (B, N), "global_batch": B * world, "phonemes": N,
                   "frames_per_utt": int(frames / args.steps / world / B), "parallelism": "utterance-sharded x%d" % world,
                   "token_rate_precision": "f32", "frame_rate_precision": "%s/%s" % (eng.decoder_precision, eng.vocoder_precision)},
        "latency": lat, "roofline": roof, "stage_ms": {k: round(v, 3) for k, v in stages.items()}, "kernels_ms": kernels,
    }
    if world == 1 and args.cpu_utts > 0:
        line["cpu_baseline"] = cpu_baseline(args.cpu_utts, N)
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
