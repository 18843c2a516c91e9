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
