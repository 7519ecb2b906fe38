#!/usr/bin/env python3
"""Where a wave of k_boxcar_scan spends its cycles, by phase of the tile loop (a build with -DRTLFM_BOX_PHASES:
tools/build_variant.sh box_phases -DRTLFM_BOX_PHASES).  256 streams x 64 x 262144 B, boxcar / D, -A std | fast."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import ATAN_FAST, ATAN_STD, RtlfmCfg  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    dev = torch.device("cuda:0")
    S, NB, L = 256, 64, 262144
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=2.4e6, dev_hz=75e3, amplitude=40.0)
    lib = os.path.join(ROOT, "build_ablate", "lib_box_phases.so")
    names = ("stage + flush + loads", "running sums", "scan + prefixes to LDS", "output loop + rest")
    for D in (84, 10, 6):
        for atan in (ATAN_STD, ATAN_FAST):
            cfg = RtlfmCfg.default(downsample=D, downsample_passes=0, rate_out=int(2.4e6 / D), custom_atan=atan, block_len=L, max_blocks=NB)
            with GpuDemod(cfg, S, 0, lib_path=lib) as g:
                cap = g.result_cap(NB)
                out = torch.empty((S, cap), dtype=torch.int16, device=dev)
                n = torch.zeros(S, dtype=torch.int32, device=dev)
                for _ in range(20):
                    g.run_device(iq.data_ptr(), iq.stride(0), NB, out.data_ptr(), cap, n.data_ptr())
                g.sync()
                g.timing_enable(True); g.timing_read()
                for _ in range(20):
                    g.run_device(iq.data_ptr(), iq.stride(0), NB, out.data_ptr(), cap, n.data_ptr())
                ms, cnt = g.timing_read()
                g.timing_enable(False)
                g.clock_probe(True)
                g.run_device(iq.data_ptr(), iq.stride(0), NB, out.data_ptr(), cap, n.data_ptr())
                st = g.clock_stamps()
                g.clock_probe(False)
                tiles = S * NB * L / 8192
                tot = st.sum(axis=0).astype(np.float64)
                per_tile = tot / (tiles * (1 + 1.0 / 16))  # + warm-up tiles, roughly
                print(f"/{D} -A {'std' if atan == ATAN_STD else 'fast'}: {ms / cnt:.4f} ms per launch (unstamped), {len(st)} waves; wave cycles per tile by phase:")
                for k in range(4):
                    print(f"    {names[k]:26s} {per_tile[k]:8.0f}  ({100 * tot[k] / tot.sum():4.1f} %)")
                print(f"    {'sum':26s} {per_tile.sum():8.0f}", flush=True)


if __name__ == "__main__":
    main()
