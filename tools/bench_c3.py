#!/usr/bin/env python3
"""C3 of BASELINE.json on one GPU: 4096 NBFM streams at 1.024 MS/s, decimate-by-64 (+ FIR),
deemph, arbitrary_resample to 22.05 kHz.  Wall time per step over all kernels (fused front
end + tail), for DESIGN.md; `rocprofv3 --kernel-trace --stats` on it gives the split.

    python tools/bench_c3.py [--streams 4096] [--blocks 4] [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import RESAMPLE_ARBITRARY, RtlfmCfg, load  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--block-len", type=int, default=262144)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tail", type=int, default=1, help="0: front end only (no deemph / resampler)")
    ap.add_argument("--preset", choices=["c3", "wbfm"], default="c3",
                    help="wbfm: rtl_fm -M wbfm (boxcar /6, -A fast, deemph, low_pass_real 170k -> 32k) on 1.02 MS/s streams")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = load()
    if a.preset == "wbfm":
        from rtlsdr_amd.capi import ATAN_FAST, RESAMPLE_LOW_PASS_REAL
        kw = dict(downsample=6, downsample_passes=0, custom_atan=ATAN_FAST, rate_out=170000, block_len=a.block_len,
                  max_blocks=a.blocks)
        if a.tail:
            kw.update(deemph=1, deemph_a=lib.rtlfm_deemph_a(170000, 75), rate_out2=32000, resampler=RESAMPLE_LOW_PASS_REAL)
    else:
        kw = dict(downsample=64, downsample_passes=6, comp_fir_size=9, rate_out=16000, block_len=a.block_len,
                  max_blocks=a.blocks)
        if a.tail:
            kw.update(deemph=1, deemph_a=lib.rtlfm_deemph_a(16000, 75), rate_out2=22050, resampler=RESAMPLE_ARBITRARY)
    cfg = RtlfmCfg.default(**kw)
    S, NB, L = a.streams, a.blocks, a.block_len
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=1.024e6, dev_hz=5e3, amplitude=40.0 if a.preset == "wbfm" else 60.0)
    g = GpuDemod(cfg, S, 0)
    cap = g.result_cap(NB)
    out = torch.empty((S, cap), dtype=torch.int16, device=dev)
    out_len = torch.zeros(S, dtype=torch.int32, device=dev)

    def step():
        g.run_device(iq.data_ptr(), iq.stride(0), NB, out.data_ptr(), out.stride(0), out_len.data_ptr())

    for _ in range(a.warmup):
        step()
    g.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    g.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    nbytes = S * NB * L
    print(f"{a.preset} tail={a.tail}: {S} streams x {NB} x {L} B: {dt * 1e3:.3f} ms/step, {nbytes / 2 / dt / 1e9:.1f} GS/s, "
          f"{nbytes / dt / 1e9:.0f} GB/s of input; out_len[0]={int(out_len[0])} path={g.last_path}")


if __name__ == "__main__":
    main()
