#!/usr/bin/env python3
"""What lies between two front-end launches that follow each other?  (GPU box)

Every wave stamps the 100 MHz real-time counter at its first and last instruction (rtlfm_gpu_clock_probe); two stamped
launches in a row write into alternating halves of the stamp buffer, so the tool sees, for launches k and k + 1 of a
back-to-back run: the span of each (first wave start .. last wave end), and the GAP between them (last wave end of k ..
first wave start of k + 1) - the launch's fixed cost, which a launch of ONE buffer per stream pays four times as
often.  Beside them the period of the same run by the host's clock (N launches, one sync).

    python tools/launch_gap.py [--nb 1 4] [--waves 0 4096 8192] [--passes 4]
"""
import argparse
import os
import sys
import time

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
    ap.add_argument("--waves", type=int, nargs="+", default=[0, 4096, 8192, 16384])
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    S, L = a.streams, 262144
    D = 1 << a.passes
    iq = synth.fm_iq_u8_torch(S, max(a.nb) * L // 2, dev, fs=2.4e6, dev_hz=75e3)
    for nb in a.nb:
        for w in a.waves:
            cfg = RtlfmCfg.default(downsample=D, downsample_passes=a.passes, rate_out=int(2.4e6 / D), block_len=L, max_blocks=nb)
            opts = dict(fused_waves=w) if w else {}
            with GpuDemod(cfg, S, 0, options=opts) as g:
                cap = g.result_cap(nb)
                out = torch.empty((S, cap), dtype=torch.int16, device=dev)
                n = torch.zeros(S, dtype=torch.int32, device=dev)

                def step():
                    g.run_device(iq.data_ptr(), iq.stride(0), nb, out.data_ptr(), out.stride(0), n.data_ptr())

                def period(k):
                    g.sync()
                    t0 = time.perf_counter()
                    for _ in range(k):
                        step()
                    g.sync()
                    return (time.perf_counter() - t0) * 1e3 / k
                period(a.steps)
                plain = period(a.steps)
                g.clock_probe(True)
                stamped = period(a.steps)
                last = g.clock_stamps()
                g.set_option("fused_debug", 64)
                prev = g.clock_stamps()
                g.set_option("fused_debug", 0)
                g.clock_probe(False)
            s0, e0 = prev[:, 2].astype(np.int64), prev[:, 3].astype(np.int64)
            s1, e1 = last[:, 2].astype(np.int64), last[:, 3].astype(np.int64)
            span0, span1 = (e0.max() - s0.min()) / 100.0, (e1.max() - s1.min()) / 100.0
            gap = (s1.min() - e0.max()) / 100.0
            per = (s1.min() - s0.min()) / 100.0
            # how the first launch drains and the second ramps: waves still running / already running around the boundary
            ramp = np.sort(s1 - s1.min()) / 100.0
            drain = np.sort(e0.max() - e0) / 100.0
            print(f"nb={nb} fused_waves={w or 'plan':>5}: period {plain * 1e3:7.1f} us (stamped {stamped * 1e3:7.1f}, by the stamps {per:7.1f}) | "
                  f"span {span0:7.1f} / {span1:7.1f} us, GAP {gap:5.1f} us | waves {len(s1)}; first 4096 waves started within {ramp[min(4095, len(ramp) - 1)]:5.1f} us; "
                  f"the last 4096 / 1024 / 256 waves ended within {drain[min(4095, len(drain) - 1)]:5.1f} / {drain[min(1023, len(drain) - 1)]:5.1f} / {drain[min(255, len(drain) - 1)]:5.1f} us of the end",
                  flush=True)


if __name__ == "__main__":
    main()
