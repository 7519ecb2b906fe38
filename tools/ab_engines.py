#!/usr/bin/env python3
"""Interleaved A/B of the fused kernel's pass-0 engines on the C2 workload.

One process, one input, two handles (path 3 = v_dot4, path 4 = int8 MFMA); the
launches alternate A B A B ... so that clock/thermal drift of the box hits both
engines alike.  Per-launch time comes from the library's HIP events.

    python tools/ab_engines.py [--rounds 60] [--burst 5] [--paths 3 4]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import ATAN_FAST, ATAN_LUT, ATAN_STD, RtlfmCfg  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=64)
    ap.add_argument("--block-len", type=int, default=262144)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--fir9", action="store_true")
    ap.add_argument("--boxcar", type=int, default=0, help="D > 0: the low_pass boxcar front end instead of fifth_order passes")
    ap.add_argument("--atan", choices=["std", "fast", "lut"], default="std")
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("--burst", type=int, default=5, help="launches per engine per round")
    ap.add_argument("--paths", type=int, nargs="+", default=[3, 4])
    ap.add_argument("--waves", type=int, nargs="+", default=None, help="fused_waves per entry of --paths")
    ap.add_argument("--libs", nargs="+", default=None,
                    help="one library build per entry of --paths (default: the in-tree build for all)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    D = 1 << a.passes
    if a.boxcar:
        a.passes, a.fir9, D = 0, False, a.boxcar
    cfg = RtlfmCfg.default(downsample=D, downsample_passes=a.passes, rate_out=int(2.4e6 / D),
                           comp_fir_size=9 if a.fir9 else 0,
                           custom_atan={"std": ATAN_STD, "fast": ATAN_FAST, "lut": ATAN_LUT}[a.atan],
                           block_len=a.block_len, max_blocks=a.blocks)
    S, NB, L = a.streams, a.blocks, a.block_len
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=2.4e6, dev_hz=75e3, amplitude=40.0 if a.atan == "fast" else 60.0)
    hs = []
    libs = a.libs or [None] * len(a.paths)
    if len(libs) != len(a.paths):
        ap.error("--libs needs one entry per --paths entry")
    for path, lib in zip(a.paths, libs):
        g = GpuDemod(cfg, S, 0, lib_path=lib and os.path.abspath(lib))
        g.set_path(path)
        if a.waves:
            g.set_option("fused_waves", a.waves[len(hs)])
        hs.append(g)
    cap = hs[0].result_cap(NB)
    # the output a quarter of the HBM away from the input, as bench.py places it (rtlfm_gpu_malloc_apart)
    import ctypes as C
    far, apart = C.c_void_p(), C.c_int()
    assert hs[0].lib.rtlfm_gpu_malloc_apart(0, S * cap * 2, iq.data_ptr(), iq.numel(), C.byref(far), C.byref(apart)) == 0
    print(f"output apart from the input: {bool(apart.value)}")

    class Out:
        def data_ptr(self):
            return far.value

        def stride(self, d):
            return cap
    out = Out()
    out_len = torch.zeros(S, dtype=torch.int32, device=dev)
    ms = [[] for _ in hs]
    for g in hs:
        g.timing_enable(True)
    for r in range(a.rounds + 3):
        for i, g in enumerate(hs):
            for _ in range(a.burst):
                g.run_device(iq.data_ptr(), iq.stride(0), NB, out.data_ptr(), out.stride(0), out_len.data_ptr())
            t, n = g.timing_read()
            if r >= 3:
                ms[i].append(t / n)
    nbytes = S * NB * L
    for path, m in zip(a.paths, ms):
        m = np.array(m)
        print(f"path {path}: mean {m.mean():.4f} ms  median {np.median(m):.4f}  min {m.min():.4f}  max {m.max():.4f}"
              f"  -> {nbytes / np.median(m) / 1e6:.0f} GB/s at the median")
    if len(hs) == 2:
        r = np.array(ms[1]) / np.array(ms[0])
        print(f"P={a.passes} fir9={int(a.fir9)} atan={a.atan}: {np.median(ms[0]):.4f} {np.median(ms[1]):.4f}")
        print(f"ratio path{a.paths[1]}/path{a.paths[0]}: mean {r.mean():.4f}  median {np.median(r):.4f}")


if __name__ == "__main__":
    main()
