"""Multi-process path on CPU (gloo, world_size 2): rank 0 packs the weights, ONE broadcast ships the blob, every rank
ends up with identical bytes and a disjoint, exhaustive shard of the utterances.  (On the GPU node the same code runs
with backend "nccl" = RCCL over xGMI; there is no steady-state collective.)"""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emotivoice_amd.sharding import broadcast_blob, shard_utterances
        t = broadcast_blob(rank, world, 0, dist, dur_mode="bench", device="cpu")
        digest = hashlib.sha256(t.numpy().tobytes()).hexdigest()
        lens = [64 + (i * 7919) % 449 for i in range(37)]
        mine = shard_utterances(lens, world)[rank]
        # counters for the scaling report: frames processed per rank, summed with one all_reduce
        frames = torch.tensor([sum(lens[i] for i in mine) * 4], dtype=torch.int64)
        dist.all_reduce(frames)
        np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array(mine))
        with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
            f.write("%s %d %d %d" % (digest, t.numel(), int(frames.item()), sum(lens) * 4))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_shard_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    recs = [open(tmp_path / ("r%d.txt" % r)).read().split() for r in range(world)]
    assert recs[0][0] == recs[1][0] and recs[0][1] == recs[1][1] and int(recs[0][1]) > 100_000_000
    assert recs[0][2] == recs[0][3] == recs[1][2]          # all_reduce of the per-rank frame counters == total
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert sorted(a.tolist() + b.tolist()) == list(range(37)) and not set(a) & set(b)
