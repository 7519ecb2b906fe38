import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from cases import CASES, make_cfg
from oracle import pyoracle as po
from rtlsdr_amd import synth
from test_parity_gpu import gpu_run
po.build()
for name in ["c1_boxcar10_fast", "box256_std", "usb_box8", "box7_std_lpr"]:
    ov, sig = [(o, s) for n, o, s in CASES if n == name][0]
    for L, nb, ns in [(16384, 3, 2), (8192, 5, 3)]:
        cfg = make_cfg(ov, L, nb)
        iq = synth.fm_iq_u8(ns, L // 2 * nb, seed=4242, **sig)
        want, wl, _ = po.run_batch(cfg, iq, nthreads=2)
        outs, st, used = gpu_run(cfg, iq)
        for s in range(ns):
            w = want[s, :wl[s]]
            g = outs[s]
            if len(g) != len(w):
                print(name, L, s, "LEN", len(g), len(w)); continue
            bad = np.nonzero(g != w)[0]
            print(name, L, nb, s, "n", len(w), "bad", len(bad), bad[:12], g[bad[:4]], w[bad[:4]])
