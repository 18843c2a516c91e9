"""MI355X: the multi-process path on real hardware -- two ranks (both on device 0, the box has one GPU) share the blob through one
broadcast, synthesise disjoint shards of a ragged utterance list, and together reproduce the single-process result bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_reproduce_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(HERE, "_dist_shard_worker.py"), str(tmp_path)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    from emotivoice_amd.engine import EVEngine
    from emotivoice_amd.packer import pack_state_dict
    from emotivoice_amd.synthetic import synth_inputs, synth_state_dict
    lens = [64 + (i * 7919) % 449 for i in range(12)]
    utts = synth_inputs(3, lens, [i % 2000 for i in range(12)])
    eng = EVEngine()
    eng.load_blob(*pack_state_dict(synth_state_dict(0, "parity")))
    ref = eng.synthesize(utts)
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    seen = sorted(int(i) for p in parts for i in p["idx"])
    assert seen == list(range(12)) and not set(parts[0]["idx"]) & set(parts[1]["idx"])
    for p in parts:
        assert int(p["total_frames"]) == int(ref["mel_lens"].sum())
        for i in p["idx"]:
            assert np.array_equal(p["wav%d" % int(i)], ref["wav_list"][int(i)]), int(i)
    eng.close()


def test_bench_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` (no launcher) on the one-GPU box: both ranks on device 0 over gloo; the line must say n_gpus 2,
    both ranks seen, and twice the frames of one rank's batch."""
    import json
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
           "--force-device", "0", "--batch", "4", "--sub-batches", "1", "--phonemes", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["config"]["global_batch"] == 8
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 2 * 4 * 64 * 4) < 1.0       # frames per step over both ranks


_RCCL_WORKER = r'''
import hashlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from emotivoice_amd.engine import EVEngine
from emotivoice_amd.sharding import broadcast_blob
from emotivoice_amd.synthetic import synth_inputs
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = broadcast_blob(0, 1, 0, dist, dur_mode="parity", collective=True)    # the nccl branch: device tensors, RCCL broadcast kernels
assert t.is_cuda and t.dtype == torch.uint8
x = torch.full((1024,), 3.0, device="cuda"); dist.all_reduce(x); assert float(x.sum()) == 3072.0
eng = EVEngine(device_id=0, precision="mx")
eng.load_blob_device(t.data_ptr(), t.numel(), keepalive=t)
out = eng.synthesize(synth_inputs(3, [40], [5]))
assert np.isfinite(out["wav"]).all() and out["wav"].shape[0] == 256 * int(out["mel_lens"][0])
print("RCCL_OK", hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16], dist.get_backend())
dist.barrier(); dist.destroy_process_group()
'''


def test_rccl_backend_executes_the_blob_broadcast(tmp_path):
    """The `nccl` (= RCCL) branch of the multi-GPU path on the one GPU this box has: a world-size-1 process group on the RCCL backend,
    the packed-weight blob through sharding.broadcast_blob as a DEVICE tensor (communicator creation + the broadcast / all-reduce kernels
    really run), borrowed in place by an engine that then synthesises.  (Rounds 1-2 only ever ran the gloo branch.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import hashlib
    from emotivoice_amd.packer import pack_state_dict
    from emotivoice_amd.synthetic import synth_state_dict
    w = tmp_path / "rccl_worker.py"
    w.write_text(_RCCL_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, str(w), os.path.dirname(HERE), "29561"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_OK")][-1].split()
    blob, _ = pack_state_dict(synth_state_dict(0, "parity"))
    assert line[1] == hashlib.sha256(bytes(blob)).hexdigest()[:16] and line[2] == "nccl"
