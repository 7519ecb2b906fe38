#!/usr/bin/env python3
"""Is the launch time of the headline kernel bimodal over TIME or over BUFFERS?  (GPU box)

Three buffer sets measured round-robin for ~25 s (100 launches per sample); half way the library's
bandwidth probe (hipMalloc / hipFree of 4 GiB) runs a few times and two more sets are allocated.
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
    opts = dict(fused_gss=int(os.environ.get("GSS", "0")))
    with GpuDemod(cfg, S, 0, options=opts) as g:
        cap = g.result_cap(NB)

        def new_set(src=None):
            iq = iq0 if src is None else torch.empty_like(iq0)
            if src is not None:
                iq.copy_(src)
            return iq, torch.empty((S, cap), dtype=torch.int16, device=dev), torch.zeros(S, dtype=torch.int32, device=dev)
        sets = [new_set(), new_set(iq0), new_set(iq0)]

        def measure(st, steps=100):
            iq, out, n = st
            g.timing_enable(True); g.timing_read()
            for _ in range(steps):
                g.run_device(iq.data_ptr(), iq.stride(0), NB, out.data_ptr(), out.stride(0), n.data_ptr())
            ms, cnt = g.timing_read()
            g.timing_enable(False)
            return ms / cnt
        for st in sets:
            measure(st, 150)
        t0 = time.time()
        for it in range(36):
            if it == 18:
                rd, rw, wf = C.c_double(), C.c_double(), C.c_double()
                for _ in range(3):
                    lib.rtlfm_gpu_bw_probe(0, 4 << 30, 16, 20, C.byref(rd), C.byref(rw), C.byref(wf))
                print(f"  -- bw probe x3: read {rd.value:.0f} rw {rw.value:.0f} GB/s; two more buffer sets", flush=True)
                sets += [new_set(iq0), new_set(iq0)]
                for st in sets[3:]:
                    measure(st, 150)
            print(f"t={time.time() - t0:5.1f}s  " + "  ".join(f"{measure(st):.4f}" for st in sets), flush=True)


if __name__ == "__main__":
    main()
