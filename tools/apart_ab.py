#!/usr/bin/env python3
"""Output a quarter of the HBM away from the input, or in the same quarter: A/B per workload.  (GPU box)

For each front-end configuration the same launch on the same input with (a) an output buffer that the
placement probe puts in the input's quarter and (b) one from rtlfm_gpu_malloc_apart, interleaved.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import ATAN_FAST, ATAN_STD, RtlfmCfg, load  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = load()
    S, NB, L = 256, 64, 262144
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=2.4e6, dev_hz=75e3, amplitude=40.0)
    nbytes = iq.numel()
    cases = [("p4 /16 std", dict(downsample=16, downsample_passes=4)), ("p5 /32 std", dict(downsample=32, downsample_passes=5)),
             ("p6+fir9 /64", dict(downsample=64, downsample_passes=6, comp_fir_size=9)),
             ("p3 /8 std", dict(downsample=8, downsample_passes=3)), ("p2 /4 std", dict(downsample=4, downsample_passes=2)),
             ("box10 fast", dict(downsample=10, custom_atan=ATAN_FAST)), ("box10 std", dict(downsample=10, custom_atan=ATAN_STD)),
             ("box6 fast", dict(downsample=6, custom_atan=ATAN_FAST)), ("box16 std", dict(downsample=16)), ("box64 std", dict(downsample=64))]
    for name, ov in cases:
        cfg = RtlfmCfg.default(block_len=L, max_blocks=NB, rate_out=150000, **ov)
        with GpuDemod(cfg, S, 0) as g:
            cap = g.result_cap(NB)
            obytes = S * cap * 2
            n = torch.zeros(S, dtype=torch.int32, device=dev)
            # same quarter: try a few torch allocations until the probe says "not apart"
            near, keep = None, []
            for _ in range(6):
                t = torch.empty(obytes, dtype=torch.uint8, device=dev)
                keep.append(t)
                if lib.rtlfm_gpu_placement_probe(0, iq.data_ptr(), nbytes, t.data_ptr(), obytes, None, None) == 0:
                    near = t
                    break
            far, apart = C.c_void_p(), C.c_int()
            assert lib.rtlfm_gpu_malloc_apart(0, obytes, iq.data_ptr(), nbytes, C.byref(far), C.byref(apart)) == 0

            def measure(out_ptr, steps=100):
                g.timing_enable(True); g.timing_read()
                for _ in range(steps):
                    g.run_device(iq.data_ptr(), NB * L, NB, out_ptr, cap, n.data_ptr())
                ms, cnt = g.timing_read()
                g.timing_enable(False)
                return ms / cnt
            measure(far.value, 150)
            res = {"near": [], "far": []}
            for _ in range(4):
                if near is not None:
                    res["near"].append(measure(near.data_ptr()))
                res["far"].append(measure(far.value))
            g.sync()
            lib.rtlfm_gpu_free(far)
        D = ov["downsample"]
        alg = (2.0 + 2.0 / D) * S * NB * L / 2
        f = lambda v: f"{sum(v) / len(v):.4f} ms = {alg / (sum(v) / len(v)) / 1e6 / 8000:.3f}" if v else "n/a"  # noqa: E731
        print(f"{name:12s} same quarter: {f(res['near'])}   apart ({apart.value}): {f(res['far'])}", flush=True)


if __name__ == "__main__":
    main()
