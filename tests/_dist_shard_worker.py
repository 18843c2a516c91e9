"""Worker of tests/test_gpu_dist.py: one rank of a 2-process job on ONE GPU (both ranks on device 0, gloo rendezvous; on the 8-GPU
node the same code runs with backend nccl = RCCL and one device per rank).  Rank 0 packs the weights, one broadcast ships the
blob, every rank borrows the device blob and synthesises ITS shard of the utterances."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emotivoice_amd.engine import EVEngine
    from emotivoice_amd.sharding import broadcast_blob, shard_utterances
    from emotivoice_amd.synthetic import synth_inputs
    blob = broadcast_blob(rank, world, 0, dist, dur_mode="parity")           # CPU -> cuda:0 on every rank through gloo
    eng = EVEngine(device_id=0)
    eng.load_blob_device(blob.data_ptr(), blob.numel(), keepalive=blob)
    lens = [64 + (i * 7919) % 449 for i in range(12)]
    utts = synth_inputs(3, lens, [i % 2000 for i in range(12)])
    mine = shard_utterances(lens, world)[rank]
    out = eng.synthesize([utts[i] for i in mine])
    frames = torch.tensor([int(out["mel_lens"].sum())], dtype=torch.int64)
    dist.all_reduce(frames)                                                   # the scaling report's only steady-state collective
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), idx=np.array(mine), total_frames=int(frames.item()),
             **{"wav%d" % i: w for i, w in zip(mine, out["wav_list"])})
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
