#!/usr/bin/env python3
"""The same launch again and again: every output of every run must equal the first run's.  A rare per-wave glitch (one wrong
sample in one wave once in ten thousand launches) does not show in a parity suite of small cases; here every launch has
thousands of waves.  Shapes: the partial-tile + raw-DC-block kernel (6 passes + FIR9, buffers of 512 x 23 bytes), the headline,
config 3's front end, the boxcar front ends, 7 passes behind the emit mode."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import capi, synth  # noqa: E402
from rtlsdr_amd.capi import RtlfmCfg  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    shapes = [
        ("6 passes + FIR9, -E rdc, 11776-byte buffers", dict(downsample=64, downsample_passes=6, comp_fir_size=9, dc_block_raw=1), 11776, 5, 4096),
        ("6 passes + FIR9, -E rdc, 11776-byte buffers, one wave per stream", dict(downsample=64, downsample_passes=6, comp_fir_size=9, dc_block_raw=1), 11776, 5, 4096, dict(fused_waves=1)),
        ("4 passes, 11776-byte buffers", dict(downsample=16, downsample_passes=4), 11776, 5, 4096),
        ("4 passes -E rdc, 262144-byte buffers", dict(downsample=16, downsample_passes=4, dc_block_raw=1), 262144, 4, 1024),
        ("4 passes (headline kernel)", dict(downsample=16, downsample_passes=4), 262144, 4, 1024),
        ("boxcar /10 -A fast", dict(downsample=10, downsample_passes=0, custom_atan=1), 262144, 4, 1024),
        ("boxcar /6 -A std", dict(downsample=6, downsample_passes=0), 262144, 4, 1024),
        ("7 passes + FIR9", dict(downsample=128, downsample_passes=7, comp_fir_size=9), 262144, 4, 1024),
    ]
    for sh in shapes:
        name, ov, L, nb, S = sh[:5]
        opts = sh[5] if len(sh) > 5 else None
        cfg = RtlfmCfg.default(block_len=L, max_blocks=nb, rate_out=int(2.4e6 / ov["downsample"]), **ov)
        iq = torch.randint(0, 256, (S, nb * L), dtype=torch.uint8, device=dev)
        iq = (iq.to(torch.int16) // 4 + 96).to(torch.uint8)  # around 127, as samples are
        with GpuDemod(cfg, S, 0, options=opts) as g:
            first = None
            bad = 0
            st0 = [g.state_get(s) for s in (0, S - 1)]
            for it in range(N):
                g.reset()
                out, n = g.run_torch(iq)
                g.sync()
                if first is None:
                    first = (out.clone(), n.clone())
                    continue
                if not torch.equal(n, first[1]):
                    print(f"  run {it}: counts differ"); bad += 1; continue
                k = int(n[0].item())
                d = (out[:, :k] != first[0][:, :k])
                if d.any():
                    idx = d.nonzero()[:8].tolist()
                    print(f"  run {it}: {int(d.sum().item())} outputs differ from the first run, first at (stream, output) {idx}; per buffer {k // nb}", flush=True)
                    bad += 1
            print(f"{name}: {S} streams x {nb} x {L} B, {N} runs: {bad} runs differ from the first", flush=True)


if __name__ == "__main__":
    main()
