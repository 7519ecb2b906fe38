#!/usr/bin/env python3
"""rtl_power's everyday shapes - 2^5 ... 2^13 bins, reads of 16384 ... 65536 bytes (src/rtl_power.c:483-504: the planner
never reads less than 16384 bytes, so a read holds several frames) - through the in-LDS kernels: ms per launch and complex
samples/s for 1 GiB of IQ, next to BASELINE config 4 (2^14 bins)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd.capi import RtlpowerCfg  # noqa: E402
from rtlsdr_amd.power import GpuPower  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    total = 1 << 30
    shapes = [(14, 32768, {}), (13, 16384, {}), (12, 16384, {}), (10, 16384, {}), (8, 16384, {}), (5, 16384, {}), (12, 65536, {}), (10, 65536, {}), (13, 32768, {}),
              # a range below 1 MHz is decimated in front of the FFT (src/rtl_power.c:466-480): boxcar (default) or -F fifth_order passes
              (10, 16384, dict(downsample=8, boxcar=1)), (8, 16384, dict(downsample=4, boxcar=1)), (12, 32768, dict(downsample=4, boxcar=1)),
              (10, 16384, dict(downsample=8, downsample_passes=3, boxcar=0)), (10, 16384, dict(downsample=8, downsample_passes=3, boxcar=0, comp_fir_size=9))]
    if len(sys.argv) > 1 and sys.argv[1] == "dec":
        shapes = [x for x in shapes if x[2]]
    for bin_e, L, extra in shapes:
        streams = 1024
        nreads = total // (streams * L)
        cfg = RtlpowerCfg.default(bin_e=bin_e, window=1, buf_len=L, **extra)
        iq = torch.randint(0, 256, (streams, nreads * L), dtype=torch.uint8, device=dev)
        with GpuPower(cfg, streams, 0) as g:
            for _ in range(3):
                g.scan_device(iq.data_ptr(), iq.stride(0), nreads)
            g.sync()
            t0 = time.perf_counter()
            K = 10
            for _ in range(K):
                g.scan_device(iq.data_ptr(), iq.stride(0), nreads)
            g.sync()
            dt = (time.perf_counter() - t0) / K
        samples = streams * nreads * L // 2
        print(f"2^{bin_e} bins {extra or ''}, {streams} streams x {nreads} reads x {L} B ({L // (2 << bin_e)} frames per read): {dt * 1e3:8.3f} ms per launch, "
              f"{samples / dt / 1e9:7.2f} Gsamples/s", flush=True)
        del iq


if __name__ == "__main__":
    main()
