#!/usr/bin/env python3
"""Placement, part 2 (GPU box): is it the allocation or the address?

(a) six separate 4 GiB allocations, each measured twice (reproducible per allocation?);
(b) one 13 GiB arena, the input placed at different offsets inside it, the output inside a second arena.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import RtlfmCfg  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    S, NB, L = 256, 64, 262144
    cfg = RtlfmCfg.default(downsample=16, downsample_passes=4, rate_out=150000, block_len=L, max_blocks=NB)
    iq0 = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=2.4e6, dev_hz=75e3)
    nbytes = iq0.numel()
    with GpuDemod(cfg, S, 0) as g:
        cap = g.result_cap(NB)

        def measure(iq_ptr, out_ptr, n_ptr, steps=200):
            for _ in range(100):
                g.run_device(iq_ptr, NB * L, NB, out_ptr, cap, n_ptr)
            g.sync()
            g.timing_enable(True); g.timing_read()
            for _ in range(steps):
                g.run_device(iq_ptr, NB * L, NB, out_ptr, cap, n_ptr)
            ms, cnt = g.timing_read()
            g.timing_enable(False)
            return ms / cnt
        n = torch.zeros(S, dtype=torch.int32, device=dev)
        out0 = torch.empty((S, cap), dtype=torch.int16, device=dev)
        print("(a) separate allocations, each twice")
        bufs = [iq0]
        for r in range(5):
            b = torch.empty_like(iq0); b.copy_(iq0); bufs.append(b)
        for rep in range(2):
            print("  " + "  ".join(f"{measure(b.data_ptr(), out0.data_ptr(), n.data_ptr()):.4f}" for b in bufs), flush=True)
        outs = [torch.empty((S, cap), dtype=torch.int16, device=dev) for _ in range(4)]
        print("  same input (first), four output allocations: " + "  ".join(f"{measure(bufs[0].data_ptr(), o.data_ptr(), n.data_ptr()):.4f}" for o in outs), flush=True)
        print("  same input (last), four output allocations:  " + "  ".join(f"{measure(bufs[-1].data_ptr(), o.data_ptr(), n.data_ptr()):.4f}" for o in outs), flush=True)
        del bufs[1:], outs
        torch.cuda.empty_cache()
        print("(b) one arena, offsets")
        arena = torch.empty(13 << 30, dtype=torch.uint8, device=dev)
        oarena = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
        base, obase = arena.data_ptr(), oarena.data_ptr()
        for off in (0, 4096, 1 << 20, 2 << 20, 64 << 20, 1 << 30, (1 << 30) + (6 << 20), 4 << 30, (4 << 30) + (2 << 20), 8 << 30):
            arena[off:off + nbytes].view(S, NB * L).copy_(iq0)
            res = []
            for ooff in (0, 128 << 10, 1 << 20, 512 << 20):
                res.append(measure(base + off, obase + ooff, n.data_ptr()))
            print(f"  iq offset {off >> 20:5d} MiB (+{off % (1 << 20)} B): " + "  ".join(f"{x:.4f}" for x in res) + "   (out offsets 0, 128K, 1M, 512M)", flush=True)


if __name__ == "__main__":
    main()
