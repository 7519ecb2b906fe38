#!/usr/bin/env python3
"""rtl_power beyond one workgroup's LDS (2^15 ... 2^21 bins): ms per launch and complex samples/s
of the transform over HBM, per bin size, and the same bytes through the in-LDS kernel at 2^14 for comparison."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd.capi import RtlpowerCfg  # noqa: E402
from rtlsdr_amd.power import GpuPower  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    total = 1 << 30  # bytes per launch
    shapes = ((14, 1024), (15, 512), (17, 128), (19, 32), (20, 16), (21, 8), (17, 1), (21, 1))
    opts = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a)  # name=value: handle options (staged_pipe=0)
    only = [a for a in sys.argv[1:] if "=" not in a]
    if only:  # only these bin sizes (for a short rocprofv3 --kernel-trace --stats run)
        shapes = tuple(x for x in shapes if str(x[0]) in only and x[1] > 1)
    for bin_e, streams in shapes:
        L = 2 << bin_e
        nreads = max(1, total // (streams * L)) if streams > 1 else 16
        cfg = RtlpowerCfg.default(bin_e=bin_e, window=1, buf_len=L)
        iq = torch.randint(0, 256, (streams, nreads * L), dtype=torch.uint8, device=dev)
        with GpuPower(cfg, streams, 0) as g:
            for k, v in opts.items():
                g.set_option(k, v)
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
        print(f"2^{bin_e} bins, {streams} streams x {nreads} reads x {L} B: {dt * 1e3:8.3f} ms per launch, {samples / dt / 1e9:7.2f} Gsamples/s, {streams * nreads * L / dt / 1e9:7.1f} GB/s in", flush=True)
        del iq


if __name__ == "__main__":
    main()
