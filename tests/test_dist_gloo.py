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


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(argv, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it must BE the launcher: two ranks meet and rank 0 reports n_gpus = 2
    (round 2's bench parsed --gpus and ran one process).  --rendezvous-only stops after the process group: no GPU here."""
    r, line = _run_bench(["--gpus", "2", "--rendezvous-only"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert line == {"rendezvous_only": True, "n_gpus": 2, "ranks_seen": [0, 1]}


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r, line = _run_bench(["--gpus", "2", "--rendezvous-only"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and line is None and "WORLD_SIZE=3" in r.stderr
    r, line = _run_bench(["--gpus", "1", "--rendezvous-only"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and line is None


def test_bench_spawn_command_shape():
    import bench
    cmd = bench.spawn_command(4, ["--gpus", "4", "--steps", "3"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert os.path.basename(cmd[-5]) == "bench.py"
