#!/usr/bin/env python3
"""Which configurations still take the staged front end (path 1)?  The random configurations of the parity sweep
(tests/test_parity_gpu.py::_random_cfg) through the automatic path selection, counted by the path a run took."""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_gpu as T  # noqa: E402
from rtlsdr_amd import capi, synth  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    by = collections.Counter()
    why = collections.Counter()
    for seed in range(n):
        rng = np.random.default_rng(9000 + seed)
        ov = T._random_cfg(rng)
        L = 512 * int(rng.integers(1, 80)) if seed % 2 else int(rng.choice([8192, 16384, 16384, 32768, 4096, 24576]))
        nb, ns = 3, 2
        cfg = T.make_cfg(ov, L, nb)
        try:
            GpuDemod(cfg, ns, 0).close()
        except capi.RtlfmError:
            by["rejected"] += 1
            continue
        iq = synth.fm_iq_u8(ns, L // 2 * nb, seed=seed, fs=1.024e6, dev_hz=20e3)
        _, _, path = T.gpu_run(cfg, iq, path=0)
        by[path] += 1
        if path == 1:
            keys = tuple(sorted(k for k in ("dc_block_raw", "offset_tuning", "squelch_level", "post_downsample", "deemph", "dc_block_audio") if ov.get(k)))
            why[(ov["mode"], ov["downsample_passes"], ov["downsample"] if not ov["downsample_passes"] else 0, L, keys)] += 1
    print("runs by path (1 = staged front end, 2 = one-launch front end):", dict(by))
    for k, v in why.most_common():  # every one of them
        print(f"  {v:3d} x mode={k[0]} passes={k[1]} boxcar={k[2]} block_len={k[3]} {k[4]}")


if __name__ == "__main__":
    main()
