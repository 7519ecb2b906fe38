#!/usr/bin/env python3
"""BASELINE configs[3]: rtl_power path, 1024 streams x 2.048 MS/s, hamming window + 16k-bin
fixed-point FFT + magnitude integrate, K=64 reads of 32768 B per stream per launch (2 GiB).
Not the driver's bench (that is bench.py / configs[1]); prints one JSON line for DESIGN.md."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--reads", type=int, default=64)
    ap.add_argument("--bin-e", type=int, default=14)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    a = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge.build()
    from oracle import pyoracle as po
    from rtlsdr_amd import synth
    from rtlsdr_amd.capi import RtlpowerCfg
    from rtlsdr_amd.power import GpuPower
    L = max(16384, 2 << a.bin_e)
    cfg = RtlpowerCfg.default(bin_e=a.bin_e, window=1, buf_len=L)
    dev = torch.device("cuda", 0)
    iq = synth.fm_iq_u8_torch(a.streams, a.reads * L // 2, dev, fs=2.048e6, dev_hz=50e3)
    g = GpuPower(cfg, a.streams, 0)
    g.scan_torch(iq); g.sync(); g.clear()
    g.timing_enable(True); g.timing_read()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.scan_torch(iq)
    g.sync()
    dt = time.perf_counter() - t0
    ms, n = g.timing_read()
    samples = a.streams * a.reads * (L // 2)
    # CPU: the oracle port on all host threads (bounded)
    cores = os.cpu_count() or 1
    cs = min(a.streams, cores)
    sample = iq[:cs, :2 * L].contiguous().cpu().numpy()
    t1 = time.perf_counter(); po.power_scan_batch(cfg, sample, nthreads=cs); one = time.perf_counter() - t1
    reps = max(1, int(a.cpu_seconds / max(one, 1e-3)))
    t1 = time.perf_counter()
    for _ in range(reps): po.power_scan_batch(cfg, sample, nthreads=cs)
    cdt = time.perf_counter() - t1
    print(json.dumps({
        "workload": f"rtl_power: {a.streams} streams x {a.reads} reads x {L} B, 2^{a.bin_e}-bin fix_fft + integrate",
        "value": round(samples * a.steps / dt / 1e6, 1), "unit": "Msamples/s",
        "kernel_ms": round(ms / max(n, 1), 3),
        "hbm_GBps_algorithmic": round((2.0 * samples + 8.0 * a.streams * (1 << a.bin_e)) / (ms / max(n, 1) * 1e-3) / 1e9, 1),
        "cpu_baseline": {"value": round(reps * cs * 2 * (L // 2) / cdt / 1e6, 2), "unit": "Msamples/s", "cores": cs,
                         "kind": "port"}}))


if __name__ == "__main__":
    main()
