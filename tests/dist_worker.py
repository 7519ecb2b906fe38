"""Worker for tests/test_shard_gloo.py: one rank of a world-size-N gloo job.
The per-rank compute is the CPU oracle standing in for the GPU library (the
sharding / scatter / gather plumbing is what is under test here)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cases import CASES, make_cfg  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from rtlsdr_amd import shard, synth  # noqa: E402


def main():
    out_path, nstreams = sys.argv[1], int(sys.argv[2])
    # (optional: buffer size and buffers per stream - the world-size-8 test runs configs[4]'s 32768 streams on tiny buffers)
    L, nb = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8192, 3)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ov, sig = [(o, s) for n, o, s in CASES if n == "c2_p4_std"][0]
    cfg = make_cfg(ov, L, nb)
    iq_all = None
    if rank == 0:
        iq_all = torch.from_numpy(synth.fm_iq_u8(nstreams, L // 2 * nb, seed=2025, **sig))
    mine = shard.scatter_streams(iq_all, nstreams, L * nb)
    a, b = shard.stream_range(nstreams, world, rank)
    assert mine.shape[0] == b - a
    o, n, _ = po.run_batch(cfg, mine.numpy(), nthreads=1)
    out, lens = shard.gather_results(torch.from_numpy(o), torch.from_numpy(n), nstreams)
    # the bench's timing reduction: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    dist.barrier()
    if rank == 0:
        np.savez(out_path, out=out.numpy(), lens=lens.numpy(), iq=iq_all.numpy(), world=world,
                 ranges=np.array([shard.stream_range(nstreams, world, r) for r in range(world)]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
