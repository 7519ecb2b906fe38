#!/usr/bin/env python3
"""Does the launch time depend on WHERE the buffers are?  (GPU box)

The headline launch measured again and again in one process on freshly allocated input / output
buffers (earlier ones stay allocated, so every round lands somewhere else), and the library's own
bandwidth probe next to each.  Same bytes, same kernel: what moves is placement and time.
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth  # noqa: E402
from rtlsdr_amd.capi import RtlfmCfg, load  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    S, NB, L = 256, 64, 262144
    cfg = RtlfmCfg.default(downsample=16, downsample_passes=4, rate_out=150000, block_len=L, max_blocks=NB)
    iq0 = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=2.4e6, dev_hz=75e3)
    lib = load()
    keep = []
    with GpuDemod(cfg, S, 0) as g:
        cap = g.result_cap(NB)

        def measure(iq, out, n, steps=300):
            for _ in range(150):
                g.run_device(iq.data_ptr(), iq.stride(0), NB, out.data_ptr(), out.stride(0), n.data_ptr())
            g.sync()
            g.timing_enable(True); g.timing_read()
            for _ in range(steps):
                g.run_device(iq.data_ptr(), iq.stride(0), NB, out.data_ptr(), out.stride(0), n.data_ptr())
            ms, cnt = g.timing_read()
            g.timing_enable(False)
            return ms / cnt
        for r in range(7):
            if r == 0:
                iq = iq0
            elif r == 6:
                iq = iq0  # the first placement again: drift check
            else:
                pad = torch.empty(((r * 37) % 11 + 1) << 20, dtype=torch.uint8, device=dev)  # shift the next allocation
                iq = torch.empty_like(iq0); iq.copy_(iq0)
                keep.append(pad)
            out = torch.empty((S, cap), dtype=torch.int16, device=dev)
            n = torch.zeros(S, dtype=torch.int32, device=dev)
            keep += [iq, out, n]
            t = measure(iq, out, n)
            rd, rw, wf = C.c_double(), C.c_double(), C.c_double()
            lib.rtlfm_gpu_bw_probe(0, 4 << 30, 16, 20, C.byref(rd), C.byref(rw), C.byref(wf))
            print(f"round {r}: iq @0x{iq.data_ptr():x} (mod 1GiB {iq.data_ptr() % (1 << 30) >> 20} MiB) out @0x{out.data_ptr():x}: {t:.4f} ms = "
                  f"{4563402752 / t / 1e6 / 8000:.3f} | probe read {rd.value:.0f} rw {rw.value:.0f} GB/s", flush=True)
    # and with raw hipMalloc'ed buffers (no torch caching allocator in between)
    time.sleep(0.1)


if __name__ == "__main__":
    main()
