"""Multi-GPU plumbing: one process per GPU, utterances are independent units.

The reference's multi-GPU inference is process-per-GPU chunking where every process re-reads the
checkpoint from disk (inference_tts.py:178-220).  Here rank 0 packs the weights once and ships the
blob with ONE broadcast (RCCL over xGMI, backend "nccl" on ROCm; "gloo" in the CPU tests); there is
no steady-state collective.
"""
from __future__ import annotations

from typing import List, Sequence


def broadcast_blob(rank: int, world: int, local_rank: int, dist, dur_mode: str = "bench", state_dict=None, device=None, collective=None):
    """Return a uint8 tensor holding the packed weight blob on this rank's device (CPU if device='cpu').  ``collective``: run the two
    broadcasts (size, bytes) even in a world of one -- how the single-GPU test box exercises the RCCL branch; default: world > 1."""
    import numpy as np
    import torch

    from .packer import pack_state_dict
    from .synthetic import synth_state_dict

    dev = torch.device(device) if device is not None else torch.device("cuda", local_rank)
    if rank == 0:
        sd = state_dict if state_dict is not None else synth_state_dict(0, dur_mode)
        blob, _ = pack_state_dict(sd)
        t = torch.from_numpy(np.frombuffer(blob, np.uint8).copy())
        n = torch.tensor([t.numel()], dtype=torch.int64)
    else:
        t, n = None, torch.zeros(1, dtype=torch.int64)
    if world > 1 if collective is None else collective:
        n = n.to(dev)
        dist.broadcast(n, 0)
        if rank != 0:
            t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
        else:
            t = t.to(dev)
        dist.broadcast(t, 0)
    else:
        t = t.to(dev)
    return t


def shard_utterances(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Deal utterance indices to ranks so that ragged work balances: longest first, each to the currently
    least-loaded rank (work ~ phoneme count).  Deterministic; every index appears exactly once."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += lengths[i]
    return out
