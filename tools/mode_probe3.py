#!/usr/bin/env python3
"""Fast / slow (input, output) pairs against the segment geometry.  (GPU box)"""
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
    with GpuDemod(cfg, S, 0) as g:
        cap = g.result_cap(NB)
        n = torch.zeros(S, dtype=torch.int32, device=dev)

        def measure(iq, out, steps=100, warm=30):
            for _ in range(warm):
                g.run_device(iq.data_ptr(), NB * L, NB, out.data_ptr(), cap, n.data_ptr())
            g.sync()
            g.timing_enable(True); g.timing_read()
            for _ in range(steps):
                g.run_device(iq.data_ptr(), NB * L, NB, out.data_ptr(), cap, n.data_ptr())
            ms, cnt = g.timing_read()
            g.timing_enable(False)
            return ms / cnt
        tin, tout = [iq0], []
        for _ in range(3):
            b = torch.empty_like(iq0); b.copy_(iq0); tin.append(b)
        for _ in range(4):
            tout.append(torch.empty((S, cap), dtype=torch.int16, device=dev))
        measure(tin[0], tout[0], 100, 150)
        pairs = [(i, o) for i in range(4) for o in range(4)]
        print("pairs (in, out): " + " ".join(f"{i}{o}" for i, o in pairs))
        for label, opts in [("tps=64 (default)", dict(fused_tiles_per_seg=0, fused_gss=0)), ("tps=63", dict(fused_tiles_per_seg=63)),
                            ("tps=65", dict(fused_tiles_per_seg=65)), ("tps=60", dict(fused_tiles_per_seg=60)),
                            ("tps=48", dict(fused_tiles_per_seg=48)), ("tps=96", dict(fused_tiles_per_seg=96)),
                            ("tps=128", dict(fused_tiles_per_seg=128)), ("tps=32", dict(fused_tiles_per_seg=32)),
                            ("gss=15", dict(fused_tiles_per_seg=0, fused_gss=15)), ("gss=20 min16", dict(fused_gss=20, fused_min_tiles=16))]:
            for k, v in opts.items():
                g.set_option(k, v)
            print(f"{label:18s} " + " ".join(f"{measure(tin[i], tout[o]) * 1000:4.0f}" for i, o in pairs), flush=True)


if __name__ == "__main__":
    main()
