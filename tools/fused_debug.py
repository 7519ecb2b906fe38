"""Where does the fused kernel differ from the staged one? (GPU box only)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from cases import make_cfg
from rtlsdr_amd import synth
from test_parity_gpu import gpu_run, FUSED_CASES
import golden_util as gu
for (passes, fir9, atan, offs) in FUSED_CASES:
    for (L, nb, ns) in [(8192, 5, 3), (16384, 4, 40), (262144, 3, 2)]:
        ov = dict(downsample=1 << passes, downsample_passes=passes, comp_fir_size=9 if fir9 else 0,
                  custom_atan=atan, offset_tuning=offs)
        cfg = make_cfg(ov, L, nb)
        iq = synth.fm_iq_u8(ns, L // 2 * nb, seed=1000 + passes, amplitude=30.0 if atan == 1 else 60.0)
        fo, fs_, u2 = gpu_run(cfg, iq, path=2)
        so, ss, u1 = gpu_run(cfg, iq, path=1)
        per_tile = 4096 >> passes
        bad = []
        for s in range(ns):
            d = np.nonzero(fo[s] != so[s])[0]
            if d.size: bad.append((s, d.size, [(int(i) // per_tile, int(i) % per_tile) for i in d[:6]]))
        sbad = [s for s in range(ns) if gu.state_dict(fs_[s], False) != gu.state_dict(ss[s], False)]
        tag = "OK " if not bad and not sbad else "BAD"
        print(tag, f"P={passes} fir={fir9} atan={atan} offs={offs} L={L} nb={nb} ns={ns}", bad[:3], "state-bad", sbad[:4])
        if sbad:
            a, b = gu.state_dict(fs_[sbad[0]], False), gu.state_dict(ss[sbad[0]], False)
            for k in a:
                if a[k] != b[k]: print("    ", k, a[k], b[k])
