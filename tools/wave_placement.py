#!/usr/bin/env python3
"""Which waves of a front-end launch are slow, and where did they run?  (GPU box)

One launch of north_star's shape with fused_debug = 2 | 32: every wave records the 100 MHz counter at
its first and last instruction and, instead of the shader clock, HW_ID / XCC_ID.  Prints the wave
duration grouped by XCD, by shader engine, by CU, by SIMD and by waves-per-SIMD at start.

    python tools/wave_placement.py [--nb 4] [--waves 4096]
"""
import argparse
import os
import sys
from collections import Counter, defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import RtlfmCfg  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--nb", type=int, default=4)
    ap.add_argument("--waves", type=int, default=4096)
    ap.add_argument("--passes", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    S, L, nb = a.streams, 262144, a.nb
    D = 1 << a.passes
    iq = synth.fm_iq_u8_torch(S, nb * L // 2, dev, fs=2.4e6, dev_hz=75e3)
    cfg = RtlfmCfg.default(downsample=D, downsample_passes=a.passes, rate_out=int(2.4e6 / D), block_len=L, max_blocks=nb)
    with GpuDemod(cfg, S, 0, options=dict(fused_waves=a.waves, fused_min_tiles=4)) as g:
        cap = g.result_cap(nb)
        out = torch.empty((S, cap), dtype=torch.int16, device=dev)
        n = torch.zeros(S, dtype=torch.int32, device=dev)
        for _ in range(300):
            g.run_device(iq.data_ptr(), iq.stride(0), nb, out.data_ptr(), out.stride(0), n.data_ptr())
        g.sync()
        g.set_option("fused_debug", 2 | 16 | 32)
        for _ in range(5):
            g.run_device(iq.data_ptr(), iq.stride(0), nb, out.data_ptr(), out.stride(0), n.data_ptr())
        st = g.clock_stamps()
    hw = st[:, 0]
    rt0, rt1 = st[:, 2].astype(np.int64), st[:, 3].astype(np.int64)
    base = rt0.min()
    start, dur = (rt0 - base) / 100.0, (rt1 - rt0) / 100.0
    lo = (hw & 0xffffffff).astype(np.int64)
    xcc = ((hw >> 32) & 0xf).astype(np.int64)
    wave_id, simd, cu, sh, se = lo & 0xf, (lo >> 4) & 3, (lo >> 8) & 0xf, (lo >> 12) & 1, (lo >> 13) & 7
    print(f"{len(st)} waves, span {(rt1.max() - base) / 100.0:.1f} us, duration p5 {np.percentile(dur, 5):.1f} p50 {np.percentile(dur, 50):.1f} "
          f"p95 {np.percentile(dur, 95):.1f}; start max {start.max():.1f} us")
    first = start < 5.0  # the waves resident from the beginning

    def group(name, key):
        d = defaultdict(list)
        for k, v in zip(key[first], dur[first]):
            d[int(k)].append(v)
        print(name + ": " + "  ".join(f"{k}: n={len(v)} mean {np.mean(v):.0f}" for k, v in sorted(d.items())))
    group("by XCD", xcc)
    group("by SE ", se)
    group("by SH ", sh)
    group("by CU ", cu)
    group("by SIMD", simd)
    group("by wave slot", wave_id)
    # occupancy at start: waves per (xcc, se, sh, cu, simd)
    slot = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
    occ = Counter(slot[first].tolist())
    per = np.array([occ[int(s)] for s in slot[first]])
    for k in sorted(set(per.tolist())):
        m = per == k
        print(f"SIMDs holding {k} waves at start: {int(m.sum()) // k} SIMDs, wave duration mean {dur[first][m].mean():.0f} us")
    cuslot = xcc * 10000 + se * 1000 + sh * 100 + cu
    occ = Counter(cuslot[first].tolist())
    per = np.array([occ[int(s)] for s in cuslot[first]])
    print("CUs by resident waves at start: " + "  ".join(f"{k} waves: {int((per == k).sum()) // k} CUs, mean {dur[first][per == k].mean():.0f} us" for k in sorted(set(per.tolist()))))
    print(f"distinct CUs used: {len(occ)}")


if __name__ == "__main__":
    main()
