#!/usr/bin/env python3
"""Where does a short front-end launch lose its time?  (GPU box)

For north_star's shapes (4096 streams x NB buffers of 262144 B per launch, /16 FM chain) and a few
segmentations (rtlfm_gpu_set_option fused_waves / fused_min_tiles): HIP-event time per launch in a
back-to-back run, and from the in-kernel clock stamps of one launch (rtlfm_gpu_clock_stamps) the
spread of the waves' start and end times - launch ramp, tail skew - and the per-wave duration.

    python tools/wave_timeline.py [--nb 1 4] [--waves 4096 8192 16384] [--passes 4]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import RtlfmCfg  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--nb", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--waves", type=int, nargs="+", default=[4096, 8192, 16384])
    ap.add_argument("--min-tiles", type=int, nargs="+", default=[4])
    ap.add_argument("--gss", type=int, nargs="+", default=[0], help="fused_gss values (x10; 0 = equal segments)")
    ap.add_argument("--streams-c2", action="store_true", help="the c2 shape: 256 streams x 16*nb buffers")
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--boxcar", type=int, default=0)
    ap.add_argument("--steps", type=int, default=400)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    S, L = a.streams, 262144
    if a.streams_c2:
        S, a.nb = 256, [16 * x for x in a.nb]
    D = a.boxcar or (1 << a.passes)
    nbmax = max(a.nb)
    iq = synth.fm_iq_u8_torch(S, nbmax * L // 2, dev, fs=2.4e6, dev_hz=75e3)
    import itertools
    for nb in a.nb:
        for w, gss, mt in itertools.product(a.waves, a.gss, a.min_tiles):
            cfg = RtlfmCfg.default(downsample=D, downsample_passes=0 if a.boxcar else a.passes, rate_out=int(2.4e6 / D),
                                   block_len=L, max_blocks=nb)
            with GpuDemod(cfg, S, 0, options=dict(fused_waves=w, fused_min_tiles=mt, fused_gss=gss)) as g:
                cap = g.result_cap(nb)
                out = torch.empty((S, cap), dtype=torch.int16, device=dev)
                n = torch.zeros(S, dtype=torch.int32, device=dev)

                def step():
                    g.run_device(iq.data_ptr(), iq.stride(0), nb, out.data_ptr(), out.stride(0), n.data_ptr())
                for _ in range(a.steps):
                    step()
                g.sync()
                g.timing_enable(True); g.timing_read()
                for _ in range(a.steps):
                    step()
                ms, cnt = g.timing_read()
                g.timing_enable(False)
                g.clock_probe(True)
                for _ in range(3):
                    step()
                st = g.clock_stamps()
                g.clock_probe(False)
            launch = ms / cnt
            gbs = (2.0 + 2.0 / D) * S * nb * (L // 2) / (launch * 1e-3) / 1e9
            line = f"S={S} nb={nb} fused_waves={w:6d} gss={gss:3d} min_tiles={mt:2d}: {launch:.4f} ms/launch = {gbs:7.1f} GB/s ({gbs / 8000:.3f})"
            if st is not None and len(st):
                rt0, rt1 = st[:, 2].astype(np.int64), st[:, 3].astype(np.int64)
                base = rt0.min()
                s_us, e_us = (rt0 - base) / 100.0, (rt1 - base) / 100.0
                dur = e_us - s_us
                mhz = (st[:, 1] - st[:, 0]).astype(np.float64) / np.maximum(rt1 - rt0, 1) * 100.0
                q = lambda v, p: float(np.percentile(v, p))  # noqa: E731
                line += (f" | waves {len(st)} ({len(st) // S}/stream) span {e_us.max():.1f} us; start p50 {q(s_us, 50):.1f} p99 {q(s_us, 99):.1f} max {s_us.max():.1f};"
                         f" end p1 {q(e_us, 1):.1f} p50 {q(e_us, 50):.1f}; dur p5 {q(dur, 5):.1f} p50 {q(dur, 50):.1f} p95 {q(dur, 95):.1f}; {mhz.mean():.0f} MHz")
            print(line, flush=True)


if __name__ == "__main__":
    main()
