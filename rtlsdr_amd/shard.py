"""Stream → GPU sharding.

rtl_fm's chain touches only its own ``demod_state`` (reference
src/rtl_fm.c:1179-1272): streams are independent, buffers of one stream are
ordered.  So a batch is sharded by STREAM INDEX in contiguous ranges, one
process per GPU, with no collective on the data path.  The only optional
exchanges are the ones below: handing each rank its streams when all IQ
arrives at one rank, and collecting the int16 audio (≤ 1/16 of the input bytes).
Both work on gloo/CPU tensors and on nccl(=RCCL)/GPU tensors.
"""
from __future__ import annotations


def stream_range(nstreams: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous [first, last) of the streams rank `rank` owns; sizes differ by at most 1."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(nstreams, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def owner_of(stream: int, nstreams: int, world: int) -> int:
    for r in range(world):
        a, b = stream_range(nstreams, world, r)
        if a <= stream < b:
            return r
    raise ValueError("stream out of range")


def scatter_streams(iq_root, nstreams: int, bytes_per_stream: int, device=None, src: int = 0, group=None):
    """Rank `src` holds uint8 [nstreams, bytes_per_stream]; every rank returns its
    own [n_local, bytes_per_stream] shard.  Over RCCL each peer is reached over its
    own xGMI link (grouped send/recv underneath), so the cost is one shard per
    link, not the whole batch over one link."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    a, b = stream_range(nstreams, world, rank)
    mine = torch.empty((b - a, bytes_per_stream), dtype=torch.uint8, device=device)
    if rank == src:
        reqs = []
        for r in range(world):
            ra, rb = stream_range(nstreams, world, r)
            part = iq_root[ra:rb]
            if r == src:
                mine.copy_(part)
            elif rb > ra:
                reqs.append(dist.isend(part.contiguous(), dst=r, group=group))
        for q in reqs:
            q.wait()
    elif b > a:
        dist.recv(mine, src=src, group=group)
    return mine


def gather_results(out_local, len_local, nstreams: int, dst: int = 0, group=None):
    """Collect per-stream int16 results ([n_local, cap] + int32 [n_local]) on `dst`.
    Returns (out [nstreams, cap], out_len [nstreams]) on dst, (None, None) elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cap = out_local.shape[1]
    if rank == dst:
        out = torch.empty((nstreams, cap), dtype=out_local.dtype, device=out_local.device)
        lens = torch.empty(nstreams, dtype=len_local.dtype, device=len_local.device)
        for r in range(world):
            a, b = stream_range(nstreams, world, r)
            if r == dst:
                out[a:b].copy_(out_local)
                lens[a:b].copy_(len_local)
            elif b > a:
                dist.recv(out[a:b], src=r, group=group)
                dist.recv(lens[a:b], src=r, group=group)
        return out, lens
    if out_local.shape[0] > 0:
        dist.send(out_local.contiguous(), dst=dst, group=group)
        dist.send(len_local.contiguous(), dst=dst, group=group)
    return None, None
