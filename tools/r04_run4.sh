#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "any_512n or not_powers or boxcar or segments_inside or short_callback or golden" > $OUT/gpu_tests_w.txt 2>&1
tail -15 $OUT/gpu_tests_w.txt
# ns4096 x 1 buffer: one wave per stream (no warm-up tile) against two
python3 - > $OUT/x1_waves.txt 2>&1 <<'PY'
import sys, time, ctypes as C
sys.path.insert(0, '.')
import torch
from rtlsdr_amd import synth
from rtlsdr_amd.capi import RtlfmCfg
from rtlsdr_amd.demod import GpuDemod
dev = torch.device('cuda:0')
S, L = 4096, 262144
iq = synth.fm_iq_u8_torch(S, 4 * L // 2, dev, fs=2.4e6, dev_hz=75e3, amplitude=60.0)
for nb in (1, 4):
    cfg = RtlfmCfg.default(downsample=16, downsample_passes=4, rate_out=150000, block_len=L, max_blocks=nb)
    alg = 2.125 * S * nb * L // 2
    gs = {}
    for waves in (4096, 8192, 12288, 16384):
        g = GpuDemod(cfg, S, 0, options=dict(fused_waves=waves))
        cap = g.result_cap(nb)
        far, apart = C.c_void_p(), C.c_int()
        assert g.lib.rtlfm_gpu_malloc_apart(0, S * cap * 2, iq.data_ptr(), iq.numel(), C.byref(far), C.byref(apart)) == 0
        n = torch.zeros(S, dtype=torch.int32, device=dev)
        gs[waves] = (g, far.value, cap, n)
    res = {w: [] for w in gs}
    for rnd in range(6):
        for w, (g, out, cap, n) in gs.items():
            for _ in range(100):
                g.run_device(iq.data_ptr(), iq.stride(0), nb, out, cap, n.data_ptr())
            g.sync(); g.timing_enable(True); g.timing_read()
            for _ in range(300 if nb == 1 else 100):
                g.run_device(iq.data_ptr(), iq.stride(0), nb, out, cap, n.data_ptr())
            ms, cnt = g.timing_read(); g.timing_enable(False)
            if rnd: res[w].append(ms / cnt)
    for w in gs:
        m = sum(res[w]) / len(res[w])
        print(f"nb={nb} fused_waves={w}: launch {m:.4f} ms  frac {alg / (m * 1e-3) / 8e12:.4f}   rounds {[round(x, 4) for x in res[w]]}")
    for w, (g, out, cap, n) in gs.items():
        g.lib.rtlfm_gpu_free(out); g.close()
PY
grep -v amdgpu.ids $OUT/x1_waves.txt
