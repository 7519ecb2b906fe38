"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, exports
every symbol include/rtlfm_hip.h declares, and its host-only planner helpers
agree with the oracle.  No compute entry point is called (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from rtlsdr_amd import build as hipbuild
from rtlsdr_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    hipbuild.build()
    return capi.load()


def header_symbols():
    out = []
    for hdr, prefix in (("rtlfm_hip.h", "rtlfm_"), ("rtlpower_hip.h", "rtlpower_")):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out += re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, text)
    return sorted(set(out))


def test_every_declared_symbol_is_exported(lib):
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rtlfm_hip.h but not exported"
    assert sorted(capi.DECLARED_SYMBOLS) == declared


def test_struct_sizes_match_header(lib):
    # sizeof() as the C compiler lays them out
    import subprocess
    import tempfile
    src = ('#include <stdio.h>\n#include "rtlfm_hip.h"\n#include "rtlpower_hip.h"\n'
           'int main(){printf("%zu %zu %zu\\n", sizeof(rtlfm_cfg), sizeof(rtlfm_stream_state), sizeof(rtlpower_cfg));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "s.c")
        open(p, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), p, "-o", exe])
        a, b, c = map(int, subprocess.check_output([exe]).split())
    assert a == C.sizeof(capi.RtlfmCfg)
    assert b == C.sizeof(capi.RtlfmStreamState)
    assert c == C.sizeof(capi.RtlpowerCfg)


def test_cfg_default_is_demod_init(lib):
    c = capi.RtlfmCfg()
    lib.rtlfm_cfg_default(C.byref(c))
    assert c.as_dict() == capi.RtlfmCfg.default().as_dict()


def test_planner_matches_oracle(lib, oracle_lib):
    import golden_util as gu
    for row in gu.manifest()["_optimal_settings"]:
        cfg = capi.RtlfmCfg.default(mode=row["mode"])
        cf, cr = C.c_uint32(), C.c_uint32()
        assert lib.rtlfm_optimal_settings(C.byref(cfg), row["freq"], row["rate_in"], row["min_capture_rate"],
                                          row["use_fifth_order"], 0, C.byref(cf), C.byref(cr)) == 0
        assert (cfg.downsample, cfg.downsample_passes, cfg.output_scale, cf.value, cr.value) == (
            row["downsample"], row["downsample_passes"], row["output_scale"],
            row["capture_freq"], row["capture_rate"])
    for rate in (170000, 240000, 150000, 32000, 24000, 16000):
        for tc in (75, 50):
            assert lib.rtlfm_deemph_a(rate, tc) == oracle_lib.oracle().orc_deemph_a(rate, tc)


def test_planner_random_against_live_reference(lib, oracle_lib):
    """rtlfm_optimal_settings / rtlfm_deemph_a against the reference's own static optimal_settings()
    and deemph_a expression, compiled in place, on a seeded random sweep of tool arguments."""
    if not oracle_lib.have_reference():
        pytest.skip("oracle/_ref not built here")
    import numpy as np
    rng = np.random.default_rng(1234)
    ref = oracle_lib.Reference()
    try:
        for _ in range(400):
            mode = int(rng.integers(0, 5))
            freq = int(rng.integers(24_000_000, 1_700_000_000))
            rate_in = int(rng.choice([8000, 12000, 16000, 24000, 32000, 48000, 150000, 170000, 240000, 250000,
                                      int(rng.integers(3000, 1_200_000))]))
            mcr = int(rng.choice([1_000_000, 1_300_000, 2_200_000, int(rng.integers(900_000, 3_000_000))]))
            fifth = int(rng.integers(0, 2))
            edge = int(rng.integers(0, 2))
            offs = int(rng.integers(0, 2))
            want = ref.optimal_settings(freq, rate_in, mcr, fifth, edge, mode, offs)
            cfg = capi.RtlfmCfg.default(mode=mode, offset_tuning=offs)
            cf, cr = C.c_uint32(), C.c_uint32()
            assert lib.rtlfm_optimal_settings(C.byref(cfg), freq, rate_in, mcr, fifth, edge, C.byref(cf), C.byref(cr)) == 0
            got = dict(downsample=cfg.downsample, downsample_passes=cfg.downsample_passes,
                       output_scale=cfg.output_scale, capture_freq=cf.value, capture_rate=cr.value)
            assert got == want, (mode, freq, rate_in, mcr, fifth, edge, offs)
    finally:
        ref.close()
    for rate in rng.integers(4000, 400000, 300):
        for tc in (75, 50, 25, 10):
            assert lib.rtlfm_deemph_a(int(rate), tc) == oracle_lib.oracle().orc_deemph_a(int(rate), tc)


def test_result_len_and_cap(lib, oracle_lib):
    from cases import CASES, make_cfg
    for name, ov, _ in CASES:
        for L in (512, 16384, 262144):
            cfg = make_cfg(ov, L)
            cap = lib.rtlfm_result_cap(C.byref(cfg))
            assert cap > 0 and cap <= oracle_lib.result_cap(cfg) + 2, name


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    cfg = capi.RtlfmCfg.default()
    assert lib.rtlfm_gpu_create(C.byref(cfg), 1, 0, C.byref(h)) == -19  # -ENODEV, no fallback
    assert b"no CPU fallback" in lib.rtlfm_gpu_strerror(-19)


def test_device_memory_helpers_without_gpu_fail_loudly(lib):
    """The placement and device-memory entry points (rtlfm_place.hip) say -ENODEV / -EINVAL without a device - no fallback, no
    host memory dressed as device memory - and reject bad arguments before they look for one."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_parity_gpu.py covers these")
    a, b = C.c_void_p(), C.c_void_p()
    ap, tr = C.c_int(), C.c_int()
    assert lib.rtlfm_gpu_place_pair(0, 1 << 20, 1 << 20, 1 << 30, 0, C.byref(a), C.byref(b), C.byref(ap), C.byref(tr), None, None) == -22  # max_tries < 1
    assert lib.rtlfm_gpu_place_pair(0, 0, 1 << 20, 1 << 30, 2, C.byref(a), C.byref(b), None, None, None, None) == -22
    assert lib.rtlfm_gpu_place_pair(0, 1 << 20, 1 << 20, 1 << 30, 2, C.byref(a), C.byref(b), C.byref(ap), C.byref(tr), None, None) == -19
    assert a.value is None and b.value is None
    assert lib.rtlfm_gpu_malloc(0, 4096, C.byref(a)) < 0 and a.value is None
    assert lib.rtlfm_gpu_copy(0, None, None, 16) == -22
    assert lib.rtlfm_gpu_copy(0, 16, 32, 16) == -19


def test_product_does_not_touch_oracle():
    """No file of the shipped package may reference the checker."""
    pkg = os.path.join(ROOT, "rtlsdr_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in t and "liboracle" not in t and "rtlfm_oracle" not in t, f
                assert "/root/reference" not in t, f
