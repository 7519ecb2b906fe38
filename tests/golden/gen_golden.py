#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference to build oracle/_ref):

    python tests/golden/gen_golden.py

Each fixture holds the uint8 IQ input of one stream (consecutive callback
buffers), the configuration, and what the reference's own rtlsdr_callback() +
full_demod() (compiled in place into oracle/_ref/libref_rtlfm.so) produced:
the concatenated int16 output and the carried demod_state fields.  Fixtures
are data only — no reference source text.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cases import CASES, make_cfg  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from rtlsdr_amd import synth  # noqa: E402

# (block_len, nblocks) per fixture family
SHAPES_MAIN = [(16384, 3)]
SHAPES_SMALL = [(2048, 5)]
BIG = {"c2_p4_std": [(262144, 2)], "c3_p6_fir9_deemph": [(262144, 2)], "box84_fm_squelch50": [(32768, 8)]}
MAIN = {"box42_dc", "box84_am_dc", "box6_wbfm_dc", "box334_usb", "box42_dc_arb_up32000",
        "box10_deemph_arb_down96000",
        "c1_boxcar10_fast", "c2_p4_std", "c2_p4_fir9_std", "c3_p6_fir9_deemph",
        "c3_p6_fir9_deemph_up22050", "wbfm_preset", "c2_p4_lut", "c2_p4_fast_a40",
        "box84_fm_squelch50", "raw_box10", "box10_rdc_fast"}


def main():
    po.build()
    assert po.have_reference(), "oracle/_ref not built (no /root/reference?)"
    manifest = {}
    # --only name[,name...]: add fixtures for these cases to the existing manifest (the others stay
    # byte-for-byte what earlier rounds committed)
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest = json.load(f)
    for name, ov, sig in CASES:
        if only is not None and name not in only:
            continue
        shapes = (SHAPES_MAIN if name in MAIN else SHAPES_SMALL) + BIG.get(name, [])
        for L, nb in shapes:
            cfg = make_cfg(ov, L)
            seed = synth.SEED_BASE + (zlib_crc(name) % 1000)
            iq = synth.fm_iq_u8(1, L // 2 * nb, seed=seed, **sig)[0]
            ref = po.Reference()
            out, st = ref.run_stream(cfg, iq)
            ref.close()
            fn = f"{name}_L{L}x{nb}.npz"
            np.savez_compressed(
                os.path.join(HERE, fn), iq=iq, out=out,
                cfg=np.frombuffer(bytes(cfg), dtype=np.uint8),
                state=np.frombuffer(bytes(st), dtype=np.uint8))
            manifest[fn] = dict(case=name, block_len=L, nblocks=nb, seed=seed, signal=sig,
                                cfg=cfg.as_dict(), out_len=int(out.size),
                                out_sha256=hashlib.sha256(out.tobytes()).hexdigest())
            print(fn, out.size)
    # full-scale random bytes: integer stages only (raw mode = decimator output)
    for passes in (1, 3, 6, 7) if only is None else ():
        ov = dict(mode=4, downsample=1 << passes, downsample_passes=passes, comp_fir_size=9)
        L, nb = 4096, 4
        cfg = make_cfg(ov, L)
        iq = synth.random_u8(1, L * nb, seed=77 + passes)[0]
        ref = po.Reference()
        out, st = ref.run_stream(cfg, iq)
        ref.close()
        fn = f"fullscale_raw_p{passes}_L{L}x{nb}.npz"
        np.savez_compressed(os.path.join(HERE, fn), iq=iq, out=out,
                            cfg=np.frombuffer(bytes(cfg), dtype=np.uint8),
                            state=np.frombuffer(bytes(st), dtype=np.uint8))
        manifest[fn] = dict(case="fullscale_raw", block_len=L, nblocks=nb, seed=77 + passes,
                            cfg=cfg.as_dict(), out_len=int(out.size),
                            out_sha256=hashlib.sha256(out.tobytes()).hexdigest())
        print(fn, out.size)
    if only is not None:
        with open(os.path.join(HERE, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    # planner known answers from the reference's static optimal_settings()
    ref = po.Reference()
    plan = []
    for rate_in, mcr, fifth, mode in [(240000, 2200000, 0, 0), (150000, 1300000, 1, 0),
                                      (16000, 1000000, 1, 0), (170000, 1000000, 0, 0),
                                      (24000, 1000000, 0, 1), (12000, 1000000, 1, 2),
                                      (8000, 2400000, 1, 3), (1000000, 1000000, 0, 1)]:
        r = ref.optimal_settings(100000000, rate_in, mcr, fifth, 0, mode, 0)
        plan.append(dict(freq=100000000, rate_in=rate_in, min_capture_rate=mcr,
                         use_fifth_order=fifth, mode=mode, **r))
    ref.close()
    manifest["_optimal_settings"] = plan
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


def zlib_crc(s):
    import zlib
    return zlib.crc32(s.encode())


if __name__ == "__main__":
    main()
