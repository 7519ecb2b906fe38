#!/usr/bin/env python3
"""A/B of the 16k-bin rtl_power kernel with tools/power_pair.patch applied (git apply tools/power_pair.patch, rebuild):
two 512-thread workgroups per CU ("pair" = 1) against one of 1024,
on BASELINE configs[3]'s shape (1024 streams x 64 reads x 32768 B), HIP-event time per launch."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd import synth
from rtlsdr_amd.capi import RtlpowerCfg
from rtlsdr_amd.power import GpuPower

S, NR, L = 1024, 64, 32768
cfg = RtlpowerCfg.default(bin_e=14, window=1, buf_len=L)
iq = torch.randint(0, 256, (S, NR * L), dtype=torch.uint8, device="cuda")
for pair in (1, 0, 1, 0):
    for groups in (0,):
        with GpuPower(cfg, S, 0) as g:
            g.set_option("pair", pair)
            g.set_option("groups", groups)
            for _ in range(5):
                g.scan_torch(iq)
            g.sync()
            g.timing_enable(True)
            for _ in range(20):
                g.scan_torch(iq)
            ms, n = g.timing_read()
            print(f"pair={pair} groups={groups}: {ms / n:.4f} ms per launch = {S * NR * L / 2 / (ms / n) / 1e6:.1f} G samples/s")
