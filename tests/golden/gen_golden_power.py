#!/usr/bin/env python3
"""Generate tests/golden/power_*.npz from the reference's rtl_power DSP
(oracle/_ref/libref_rtlpower.so, compiled in place from /root/reference).
Fixtures are data only: input bytes, configuration, reference avg[]/samples."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import RtlpowerCfg  # noqa: E402

CASES = [
    ("rms", dict(bin_e=0, buf_len=16384)),
    ("b4_rect", dict(bin_e=4, window=0, buf_len=16384)),
    ("b8_hamming", dict(bin_e=8, window=1, buf_len=16384)),
    ("b10_blackman", dict(bin_e=10, window=2, buf_len=16384)),
    ("b10_bh_peak", dict(bin_e=10, window=3, peak_hold=1, buf_len=16384)),
    ("b9_hannpoisson_box4", dict(bin_e=9, window=4, downsample=4, boxcar=1, buf_len=16384)),
    ("b9_youssef_fifth2_fir9", dict(bin_e=9, window=5, downsample=4, downsample_passes=2, boxcar=0,
                                   comp_fir_size=9, buf_len=16384)),
    ("b9_bartlett_fifth1", dict(bin_e=9, window=7, downsample=2, downsample_passes=1, boxcar=0, buf_len=16384)),
    ("b12_hamming", dict(bin_e=12, window=1, buf_len=16384)),
    ("c4_b14_hamming", dict(bin_e=14, window=1, buf_len=32768)),
    # beyond one workgroup's LDS: eight frames of 4096 points per read, and frequency_range()'s fine-bin plans
    ("b12_8frames_hamming", dict(bin_e=12, window=1, buf_len=65536)),
    ("b15_blackman", dict(bin_e=15, window=2, buf_len=65536)),
    ("b17_hamming", dict(bin_e=17, window=1, buf_len=262144)),
]
READS = {"b17_hamming": 2}


def main():
    po.build()
    assert po.have_power_reference()
    man = {}
    for name, kw in CASES:
        cfg = RtlpowerCfg.default(**kw)
        L, nr = int(cfg.buf_len), READS.get(name, 3)
        for kind in ("tone", "fullscale"):
            if kind == "tone":
                iq = synth.fm_iq_u8(1, L // 2 * nr, fs=2.048e6, dev_hz=20e3, seed=11)[0]
            else:
                iq = synth.random_u8(1, L * nr, seed=12)[0]
            ref = po.PowerReference()
            avg, n = ref.scan_stream(cfg, iq)
            ref.close()
            fn = f"power_{name}_{kind}.npz"
            np.savez_compressed(os.path.join(HERE, fn), iq=iq, avg=avg, samples=np.int32(n),
                                cfg=np.frombuffer(bytes(cfg), dtype=np.uint8))
            man[fn] = dict(cfg=cfg.as_dict(), reads=nr, kind=kind, samples=int(n), avg_sum=int(avg.sum()))
            print(fn, n, int(avg.sum()))
    json.dump(man, open(os.path.join(HERE, "manifest_power.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
