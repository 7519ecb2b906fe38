#!/usr/bin/env python3
"""Where a wave of the front ends spends its cycles, by phase of the tile loop (a build with both stamp sets:
tools/build_variant.sh phases -DRTLFM_BOX_PHASES -DRTLFM_FUSED_PHASES).  256 streams x 64 x 262144 B: k_fused with 4 / 5 / 6
fifth_order passes, k_boxcar_scan / D with -A std | fast."""
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
    lib = os.path.join(ROOT, "build_ablate", "lib_phases.so")
    names = ("stage + flush + loads", "running sums", "scan + prefixes to LDS", "output loop + rest")
    shapes = [(1 << P, P, ATAN_STD, fir) for P, fir in ((4, 0), (5, 0), (6, 9))] + [(D, 0, atan, 0) for D in (84, 10, 6) for atan in (ATAN_STD, ATAN_FAST)]
    fnames = ("tile arrived + staged", "pass 0 (MFMA)", "other passes (+ FIR)", "discriminator + rest")
    if True:
        for D, P, atan, fir in shapes:
            cfg = RtlfmCfg.default(downsample=D, downsample_passes=P, rate_out=int(2.4e6 / D), custom_atan=atan, block_len=L, max_blocks=NB, comp_fir_size=fir)
            nm = fnames if P else names
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
                print(f"{'fifth_order x ' + str(P) + (' + FIR9' if fir else '') if P else 'boxcar'} /{D} -A {'std' if atan == ATAN_STD else 'fast'}: {ms / cnt:.4f} ms per launch (unstamped), {len(st)} waves; wave cycles per tile by phase:")
                for k in range(4):
                    print(f"    {nm[k]:26s} {per_tile[k]:8.0f}  ({100 * tot[k] / tot.sum():4.1f} %)")
                print(f"    {'sum':26s} {per_tile.sum():8.0f}", flush=True)


if __name__ == "__main__":
    main()
