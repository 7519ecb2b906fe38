#!/usr/bin/env python3
"""rtl_fm's everyday command lines through rtlfm_gpu_run_device: ms per 4 GiB step (256 streams x 64 x 262144 B resident in HBM,
the output placed apart), which path the run took, and the fraction of the 8 TB/s peak its algorithmic bytes reach.
The planner's numbers are the reference's (rtlfm_optimal_settings, rtlfm_deemph_a)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import capi, synth  # noqa: E402
from rtlsdr_amd.capi import RtlfmCfg  # noqa: E402
from rtlsdr_amd.demod import GpuDemod  # noqa: E402

LINES = [
    ("-M fm -s 24k", dict(rate=24000)),
    ("-M fm -s 24k -E deemp", dict(rate=24000, deemp=1)),
    ("-M fm -s 24k -E deemp -E dc", dict(rate=24000, deemp=1, adc=1)),
    ("-M fm -s 12k -l 50", dict(rate=12000, squelch=50)),
    ("-M fm -s 12k -F 9", dict(rate=12000, fir=1)),
    ("-M fm -s 12k -F 9 -l 50 -E deemp", dict(rate=12000, fir=1, squelch=50, deemp=1)),
    ("-M fm -s 3k -F 9", dict(rate=3000, fir=1)),
    ("-M am -s 12k -E dc", dict(rate=12000, mode=capi.MODE_AM, adc=1)),
    ("-M usb -s 3k", dict(rate=3000, mode=capi.MODE_USB)),
    ("-M fm -s 48k -r 24k -E deemp", dict(rate=48000, deemp=1, rate2=24000)),
    ("-M wbfm", dict(rate=170000, deemp=1, rate2=32000, atan=1)),
    ("-M fm -s 240k -m 2.2M -A fast (config 0)", dict(rate=240000, atan=1, min_rate=2200000)),
    ("-M raw -s 240k", dict(rate=240000, mode=capi.MODE_RAW)),
]


def main():
    dev = torch.device("cuda:0")
    lib = capi.load()
    S, NB, L = 256, 64, 262144
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=1.0e6, dev_hz=3e3, amplitude=20.0)
    only = sys.argv[1:] if len(sys.argv) > 1 else None
    for name, o in LINES:
        if only and not any(x in name for x in only):
            continue
        cfg = RtlfmCfg.default(block_len=L, max_blocks=NB, mode=o.get("mode", capi.MODE_FM), custom_atan=o.get("atan", 0),
                               squelch_level=o.get("squelch", 0), dc_block_audio=o.get("adc", 0))
        cap_freq, cap_rate = C.c_uint32(), C.c_uint32()
        lib.rtlfm_optimal_settings(C.byref(cfg), 100000000, o["rate"], o.get("min_rate", 1000000), o.get("fir", 0), 0, C.byref(cap_freq), C.byref(cap_rate))
        if o.get("fir"):
            cfg.comp_fir_size = 9
        cfg.rate_out = o["rate"]
        if o.get("deemp"):
            cfg.deemph = 1
            cfg.deemph_a = lib.rtlfm_deemph_a(o["rate"], 75)
        if o.get("rate2"):
            cfg.rate_out2 = o["rate2"]
            cfg.resampler = capi.RESAMPLE_LOW_PASS_REAL
        D = cfg.downsample
        with GpuDemod(cfg, S, 0) as g:
            cap = g.result_cap(NB)
            far, apart = C.c_void_p(), C.c_int()
            # (a process that has allocated and freed as much as this one may need to look further than the library's own 16 GiB)
            assert lib.rtlfm_gpu_malloc_apart_ex(0, S * cap * 2, iq.data_ptr(), iq.numel(), 96 << 30, C.byref(far), C.byref(apart), None, None) == 0
            n = torch.zeros(S, dtype=torch.int32, device=dev)
            for _ in range(30):
                g.run_device(iq.data_ptr(), iq.stride(0), NB, far.value, cap, n.data_ptr())
            g.sync()
            K = 100
            t0 = time.perf_counter()
            for _ in range(K):
                g.run_device(iq.data_ptr(), iq.stride(0), NB, far.value, cap, n.data_ptr())
            g.sync()
            ms = (time.perf_counter() - t0) / K * 1e3
            out_per_in = float(n[0].item()) / (NB * L / 2)
            alg = (2.0 + 2.0 * out_per_in) * S * NB * L / 2
            print(f"{name:44s} /{D:<4d} passes {cfg.downsample_passes}  path {g.last_path}  {ms:7.3f} ms per 4 GiB  {alg / (ms * 1e-3) / 8e12:5.3f} of 8 TB/s"
                  f"  (apart {apart.value})", flush=True)
            lib.rtlfm_gpu_free(far)


if __name__ == "__main__":
    main()
