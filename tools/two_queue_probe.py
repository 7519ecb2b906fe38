#!/usr/bin/env python3
"""Does a front-end launch's ramp and drain hide behind ANOTHER launch's steady state?

north_star's live shape is 4096 streams x ONE 262144-B buffer per launch (0.21 ms): the launch is short enough that
its first and last tenth - SIMDs filling up, SIMDs running dry - cost 4-5 % against the 4-buffer launch.  Consecutive
launches of one handle cannot overlap (launch k + 1 reads the state launch k leaves), but two handles that own disjoint
halves of the streams can: each on its own HIP stream, a handle's next launch queued right behind its previous one.

    python tools/two_queue_probe.py [--nb 1] [--split 2048]

prints ms per (whole) step and the fraction of 8 TB/s for: one handle with all streams; two handles, equal split;
two handles, unequal split."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import RtlfmCfg  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--nb", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--splits", type=int, nargs="+", default=[2048, 1366, 1024])
    ap.add_argument("--steps", type=int, default=800)
    ap.add_argument("--passes", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    S, L = a.streams, 262144
    NBmax = max(a.nb)
    iq = synth.fm_iq_u8_torch(S, NBmax * L // 2, dev, fs=2.4e6, dev_hz=75e3, amplitude=60.0)
    D = 1 << a.passes
    for nb in a.nb:
        cfg = RtlfmCfg.default(downsample=D, downsample_passes=a.passes, rate_out=int(2.4e6 / D), block_len=L, max_blocks=nb)
        samples = S * nb * L // 2
        alg = (2.0 + 2.0 / D) * samples

        def run(parts):
            hs = []
            for s0, ns in parts:
                g = GpuDemod(cfg, ns, 0)
                cap = g.result_cap(nb)
                far, apart = C.c_void_p(), C.c_int()
                base = iq.data_ptr() + s0 * iq.stride(0)
                assert g.lib.rtlfm_gpu_malloc_apart(0, ns * cap * 2, base, ns * iq.stride(0), C.byref(far), C.byref(apart)) == 0
                n = torch.zeros(ns, dtype=torch.int32, device=dev)
                hs.append((g, base, far.value, cap, n))
            def step():
                for g, base, out, cap, n in hs:
                    g.run_device(base, iq.stride(0), nb, out, cap, n.data_ptr())
            for _ in range(200):
                step()
            for g, *_ in hs:
                g.sync()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            for g, *_ in hs:
                g.sync()
            dt = (time.perf_counter() - t0) / a.steps
            for g, base, out, cap, n in hs:
                g.lib.rtlfm_gpu_free(out)
                g.close()
            return dt
        one = run([(0, S)])
        print(f"nb={nb}: one handle {S} streams: {one * 1e3:.4f} ms/step  frac {alg / one / 8e12:.4f}")
        for sp in a.splits:
            two = run([(0, sp), (sp, S - sp)])
            print(f"nb={nb}: two handles {sp} + {S - sp}: {two * 1e3:.4f} ms/step  frac {alg / two / 8e12:.4f}")
        if nb == 1:
            three = run([(0, S // 4), (S // 4, S // 4), (S // 2, S // 2)])
            print(f"nb={nb}: three handles {S // 4} + {S // 4} + {S // 2}: {three * 1e3:.4f} ms/step  frac {alg / three / 8e12:.4f}")


if __name__ == "__main__":
    main()
