#!/usr/bin/env python3
"""What round 5 had left to the stage-by-stage kernels (tools/path_census.py: 7 of 561 random configurations; VERDICT r5, task
9) - `-E rdc` with offset tuning in front of the boxcar, `-E rdc` in front of the boxcar on buffers below 8192 bytes, a
boxcar beyond /2047 - each timed on 4 GiB resident in HBM beside its nearest neighbour.  Round 6 moved all three onto the
one-launch front end (profiles/r06_staged_leftovers.txt: 7.75 -> 1.54, 13.4 -> 2.5, 18.9 -> 4.1 ms; the last at 256
streams = 256 waves - beyond /2047 a run is one wave per stream).
ms per 4 GiB step, the path the run took (1 = staged, 2 = one-launch front end), fraction of 8 TB/s for the algorithmic bytes."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import capi, synth  # noqa: E402
from rtlsdr_amd.capi import RtlfmCfg  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402

# (what, cfg overrides, block_len)
LINES = [
    ("boxcar /10 -E rdc (one launch + sums pre-pass)", dict(downsample=10, dc_block_raw=1, rate_out=240000), 262144),
    ("boxcar /10 -E rdc, offset tuning (r5: staged)", dict(downsample=10, dc_block_raw=1, offset_tuning=1, rate_out=240000), 262144),
    ("boxcar /10 -E rdc, 4096-byte buffers (r5: staged)", dict(downsample=10, dc_block_raw=1, rate_out=240000), 4096),
    ("boxcar /10, 4096-byte buffers (one launch)", dict(downsample=10, rate_out=240000), 4096),
    ("boxcar /2047 (one launch)", dict(downsample=2047, rate_out=1000), 262144),
    ("boxcar /2400 (r5: staged)", dict(downsample=2400, rate_out=1000), 262144),
]


def main():
    dev = torch.device("cuda:0")
    lib = capi.load()
    total = 4 << 30
    S = 256
    iq = synth.fm_iq_u8_torch(S, total // S // 2, dev, fs=2.4e6, dev_hz=75e3, amplitude=20.0)
    for name, ov, L in LINES:
        NB = total // S // L
        cfg = RtlfmCfg.default(block_len=L, max_blocks=NB, **ov)
        with GpuDemod(cfg, S, 0) as g:
            cap = g.result_cap(NB)
            far, apart = C.c_void_p(), C.c_int()
            assert lib.rtlfm_gpu_malloc_apart_ex(0, S * cap * 2, iq.data_ptr(), iq.numel(), 96 << 30, C.byref(far), C.byref(apart), None, None) == 0
            n = torch.zeros(S, dtype=torch.int32, device=dev)
            for _ in range(10):
                g.run_device(iq.data_ptr(), iq.stride(0), NB, far.value, cap, n.data_ptr())
            g.sync()
            K = 40
            t0 = time.perf_counter()
            for _ in range(K):
                g.run_device(iq.data_ptr(), iq.stride(0), NB, far.value, cap, n.data_ptr())
            g.sync()
            ms = (time.perf_counter() - t0) / K * 1e3
            out_per_in = float(n[0].item()) / (NB * L / 2)
            alg = (2.0 + 2.0 * out_per_in) * S * NB * L / 2
            print(f"{name:52s} path {g.last_path}  {ms:8.3f} ms per 4 GiB  {alg / (ms * 1e-3) / 8e12:5.3f} of 8 TB/s", flush=True)
            lib.rtlfm_gpu_free(far)


if __name__ == "__main__":
    main()
