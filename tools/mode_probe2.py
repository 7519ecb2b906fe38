#!/usr/bin/env python3
"""Fast / slow placement (tools/mode_probe.py): which buffer, and does contiguous memory fix it?  (GPU box)"""
import ctypes as C
import os
import sys

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
    nbytes = iq0.numel()
    lib = load()
    hip = C.CDLL("libamdhip64.so")  # already loaded by torch: same runtime
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    with GpuDemod(cfg, S, 0) as g:
        cap = g.result_cap(NB)
        obytes = S * cap * 2
        n = torch.zeros(S, dtype=torch.int32, device=dev)

        def measure(iq_ptr, out_ptr, steps=100, warm=30):
            for _ in range(warm):
                g.run_device(iq_ptr, NB * L, NB, out_ptr, cap, n.data_ptr())
            g.sync()
            g.timing_enable(True); g.timing_read()
            for _ in range(steps):
                g.run_device(iq_ptr, NB * L, NB, out_ptr, cap, n.data_ptr())
            ms, cnt = g.timing_read()
            g.timing_enable(False)
            return ms / cnt
        measure(iq0.data_ptr(), torch.empty(obytes, dtype=torch.uint8, device=dev).data_ptr(), 100, 150)
        tin, tout = [iq0], []
        for _ in range(4):
            b = torch.empty_like(iq0); b.copy_(iq0); tin.append(b)
        for _ in range(5):
            tout.append(torch.empty(obytes, dtype=torch.uint8, device=dev))
        print("torch allocations: rows = input buffer, columns = output buffer")
        for i, a in enumerate(tin):
            print(f"  in{i} @0x{a.data_ptr():x}: " + "  ".join(f"{measure(a.data_ptr(), o.data_ptr()):.4f}" for o in tout), flush=True)
        cin, cout = [], []
        for _ in range(3):
            p = C.c_void_p()
            assert lib.rtlfm_gpu_malloc(0, nbytes, C.byref(p)) == 0
            hip.hipMemcpy(p, iq0.data_ptr(), nbytes, 3)
            cin.append(p.value)
            q = C.c_void_p()
            assert lib.rtlfm_gpu_malloc(0, obytes, C.byref(q)) == 0
            cout.append(q.value)
        torch.cuda.synchronize()
        print("rtlfm_gpu_malloc (contiguous): rows = input, columns = output (3 contiguous, then 2 torch)")
        for i, a in enumerate(cin):
            print(f"  cin{i} @0x{a:x}: " + "  ".join(f"{measure(a, o):.4f}" for o in cout + [t.data_ptr() for t in tout[:2]]), flush=True)
        print("torch input, contiguous output")
        for i, a in enumerate(tin[:3]):
            print(f"  in{i}: " + "  ".join(f"{measure(a.data_ptr(), o):.4f}" for o in cout), flush=True)
        for p in cin + cout:
            lib.rtlfm_gpu_free(p)


if __name__ == "__main__":
    main()
