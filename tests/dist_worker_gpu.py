"""Worker for tests/test_multigpu_gpu.py: one rank of a world-size-N job whose per-rank
compute is the HIP library (rtlfm_gpu_run_device), not the oracle.

With >= N visible GPUs: one device per rank, RCCL (backend "nccl") for the scatter of the IQ
from rank 0's GPU, the gather of the PCM and the max-reduce.  On a one-GPU box the same
control flow runs with the ranks sharing device 0 and gloo moving host tensors (RCCL
refuses two ranks on one device) — the HIP library is still what computes."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cases import CASES, make_cfg  # noqa: E402
from rtlsdr_amd import shard, synth  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    out_path, nstreams, case = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    assert ndev >= 1
    multi = ndev >= world
    local = rank if multi else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if multi:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    ov, sig = [(o, s) for n, o, s in CASES if n == case][0]
    L, nb = 16384, 3
    cfg = make_cfg(ov, L, nb)
    iq_all = None
    if rank == 0:
        iq_all = torch.from_numpy(synth.fm_iq_u8(nstreams, L // 2 * nb, seed=2026, **sig))
        if multi:
            iq_all = iq_all.to(dev)
    mine = shard.scatter_streams(iq_all, nstreams, L * nb, device=dev if multi else None)
    a, b = shard.stream_range(nstreams, world, rank)
    assert mine.shape[0] == b - a
    with GpuDemod(cfg, b - a, local) as g:
        o, n = g.run_torch(mine.to(dev))
        g.sync()
        path = g.last_path
    if not multi:
        o, n = o.cpu(), n.cpu()
    out, lens = shard.gather_results(o, n, nstreams)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev if multi else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    dist.barrier()
    if rank == 0:
        np.savez(out_path, out=out.cpu().numpy(), lens=lens.cpu().numpy(), iq=iq_all.cpu().numpy(),
                 multi=int(multi), path=path)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
