"""rtl_power: the CPU oracle against the reference's own golden output and
(where oracle/_ref exists) against the live compiled reference, function by
function."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from rtlsdr_amd.capi import RtlpowerCfg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def power_fixtures():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "power_*.npz")))


def load_power(name):
    z = np.load(os.path.join(GOLDEN, name))
    return RtlpowerCfg.from_buffer_copy(z["cfg"].tobytes()), z["iq"], z["avg"], int(z["samples"])


@pytest.mark.parametrize("name", power_fixtures())
def test_power_oracle_matches_golden(oracle_lib, name):
    cfg, iq, want, n = load_power(name)
    avg, samples = oracle_lib.power_scan_batch(cfg, iq[None, :])
    assert samples[0] == n
    assert np.array_equal(avg[0], want)  # int64, bit-exact


def test_power_functions_against_live_reference(oracle_lib):
    if not oracle_lib.have_power_reference():
        pytest.skip("oracle/_ref not built here")
    lib = oracle_lib._power_lib()
    ref = oracle_lib.PowerReference()
    rng = np.random.default_rng(9)
    try:
        for be in (3, 8, 12):
            cfg = RtlpowerCfg.default(bin_e=be, window=1, buf_len=max(16384, 2 << be))
            ref.lib.ref_power_setup(C.byref(cfg))
            n = 1 << be
            # sine table and window coefficients
            sw = np.ctypeslib.as_array(ref.lib.ref_sinewave(), shape=(n * 3 // 4,)).copy()
            mine = np.ctypeslib.as_array(C.cast(lib.orcp_sine_table(be), C.POINTER(C.c_int16)), shape=(n * 3 // 4,))
            assert np.array_equal(sw, mine)
            wc = np.ctypeslib.as_array(ref.lib.ref_window_coefs(), shape=(n,)).copy()
            assert np.array_equal(wc, oracle_lib.power_window_coefs(1, n))
            # fix_fft on full-scale data
            x = rng.integers(-32768, 32768, size=2 * n).astype(np.int16)
            a, b = x.copy(), x.copy()
            ref.lib.fix_fft(a.ctypes.data, be)
            lib.orcp_fix_fft(b.ctypes.data, be, mine.ctypes.data, be)
            assert np.array_equal(a, b)
        for length in (64, 1024, 4096):
            x = rng.integers(-3000, 3000, size=length).astype(np.int16)
            a, b = x.copy(), x.copy()
            ref.lib.fifth_order(a.ctypes.data, length); lib.orcp_fifth_order(b.ctypes.data, length)
            assert np.array_equal(a, b)
            a, b = x.copy(), x.copy()
            ref.lib.remove_dc(a.ctypes.data, length); lib.orcp_remove_dc(b.ctypes.data, length)
            assert np.array_equal(a, b)
    finally:
        ref.close()


def _random_power_cfg(rng):
    kw = dict(bin_e=int(rng.integers(0, 15)), window=int(rng.integers(0, 8)), peak_hold=int(rng.random() < 0.25))
    kw["buf_len"] = int(rng.choice([16384, 16384, 32768, 65536]))
    r = rng.random()
    if r < 0.3:
        kw.update(downsample=int(rng.choice([2, 3, 5, 7, 16])), boxcar=1)
    elif r < 0.6:
        p_ = int(rng.integers(1, 5))
        kw.update(downsample=1 << p_, downsample_passes=p_, boxcar=0, comp_fir_size=int(rng.choice([0, 9])))
    return kw


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTLFM_SWEEP_SCANNER", "64"))))
def test_oracle_scan_against_live_scanner(oracle_lib, seed):
    """The committed pin of the rtl_power oracle at the level of scanner() itself
    (src/rtl_power.c:642-720, compiled in place, reading through the file-backed device layer): seeded
    random configurations - the same generator as the GPU sweep, tests/test_power_gpu.py - FM signal and
    full-scale random bytes, 2..4 reads, avg[] (int64) and samples identical.  Configurations outside
    scanner()'s own domain (rtlpower_cfg_validate: a trailing frame past the read, where the reference
    transforms what its static buffer still holds) are skipped and counted."""
    if not oracle_lib.have_power_reference():
        pytest.skip("oracle/_ref not built here")
    from rtlsdr_amd import build as hipbuild
    from rtlsdr_amd import capi, synth
    hipbuild.build()
    lib = capi.load()
    rng = np.random.default_rng(5000 + seed)
    kw = _random_power_cfg(rng)
    cfg = RtlpowerCfg.default(**kw)
    if lib.rtlpower_cfg_validate(C.byref(cfg)) < 0:
        pytest.skip(f"outside scanner()'s domain: {kw}")
    L, nr = int(cfg.buf_len), int(rng.integers(2, 5))
    iq = np.concatenate([synth.fm_iq_u8(1, L // 2 * nr, fs=2.048e6, dev_hz=40e3, seed=100 + seed),
                         synth.random_u8(1, L * nr, seed=200 + seed)])
    avg, n = oracle_lib.power_scan_batch(cfg, iq)
    ref = oracle_lib.PowerReference()
    try:
        for s in range(2):
            ravg, rn = ref.scan_stream(cfg, iq[s])
            assert rn == n[s], (kw, s)
            assert np.array_equal(ravg, avg[s]), (kw, s)
    finally:
        ref.close()


def test_remove_dc_removes_only_half(oracle_lib):
    """p2 quirk: the sum over N/2 samples is divided by N."""
    lib = oracle_lib._power_lib()
    x = np.zeros(256, dtype=np.int16); x[0::2] = 100
    lib.orcp_remove_dc(x.ctypes.data, 256)
    assert set(x[0::2]) == {50}


def test_window_multiply_wraps(oracle_lib):
    """p3: +128 * 256 wraps to -32768 (rectangle window, full-scale input)."""
    cfg = RtlpowerCfg.default(bin_e=4, window=0, buf_len=16384)
    iq = np.full(16384, 255, dtype=np.uint8)  # 255-127 = 128 everywhere
    avg, n = oracle_lib.power_scan_batch(cfg, iq[None, :])
    assert n[0] == 16384 // 32 and avg[0].sum() > 0


PLAN_CASES = [
    # -f lower:upper:bin, crop, boxcar
    ("100M:102.048M:125", 0.0, 1), ("88M:108M:125k", 0.0, 1), ("88M:108M:1k", 0.25, 1),
    ("433M:434M:1k", 0.0, 1), ("433M:434M:1k", 0.0, 0), ("144M:146M:500", 0.1, 0),
    ("24M:1700M:1M", 0.0, 1), ("118M:137M:25k", 0.3, 1), ("100M:100.5M:50", 0.0, 1),
    ("100M:100.2M:100", 0.0, 0), ("50M:60M:2M", 0.0, 1),
]


def _atofs(t):
    mul = {"k": 1e3, "M": 1e6, "G": 1e9}.get(t[-1])
    return float(t[:-1]) * mul if mul else float(t)


@pytest.mark.parametrize("arg,crop,boxcar", PLAN_CASES)
def test_planner_and_csv_match_reference(oracle_lib, arg, crop, boxcar):
    """§8f-3: rtlpower_frequency_range / rtlpower_csv_dbm (host side of the product library)
    against the reference's frequency_range() and csv_dbm() compiled in place."""
    if not oracle_lib.have_power_reference():
        pytest.skip("oracle/_ref not built here")
    from rtlsdr_amd import capi
    from rtlsdr_amd import build as hipbuild
    hipbuild.build()
    lib = capi.load()
    lo, hi, step = (int(_atofs(x)) for x in arg.split(":"))
    plan = capi.RtlpowerPlan()
    assert lib.rtlpower_frequency_range(lo, hi, step, crop, boxcar, C.byref(plan)) == 0
    ref = oracle_lib.PowerReference()
    try:
        ref.lib.ref_frequency_range.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        ref.lib.ref_csv_dbm.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t]
        o = np.zeros(10, dtype=np.int32); rc = C.c_double()
        ref.lib.ref_frequency_range(arg.encode(), crop, boxcar, o.ctypes.data, C.byref(rc))
        last = plan.tune_count - 1
        assert (plan.tune_count, plan.rate, plan.bin_e, plan.downsample, plan.downsample_passes, plan.buf_len,
                lib.rtlpower_tune_freq(C.byref(plan), 0), lib.rtlpower_tune_freq(C.byref(plan), last), plan.crop) == \
               (o[0], o[2], o[3], o[4], o[5], o[6], o[7], o[8], rc.value)
        if plan.tune_count > 1:
            assert plan.bw_seen == o[1]
        if plan.bin_e > 14:
            return
        # csv line for hop 0 and the last hop, from the same accumulators
        rng = np.random.default_rng(3)
        avg = rng.integers(1, 1 << 40, size=1 << plan.bin_e).astype(np.int64)
        for tune in {0, last}:
            mine_avg = avg.copy()
            buf = C.create_string_buffer(1 << 20)
            n = lib.rtlpower_csv_dbm(C.byref(plan), tune, mine_avg.ctypes.data, 77, buf, len(buf))
            rbuf = C.create_string_buffer(1 << 20)
            rn = ref.lib.ref_csv_dbm(tune, avg.ctypes.data, 77, rbuf, len(rbuf))
            assert n == rn and buf.value == rbuf.value
    finally:
        ref.close()


def test_planner_random_against_live_reference(oracle_lib):
    """rtlpower_frequency_range against the reference's frequency_range() (compiled in place) on a
    seeded random sweep of ranges, bin sizes, crop factors and both decimator kinds."""
    if not oracle_lib.have_power_reference():
        pytest.skip("oracle/_ref not built here")
    from rtlsdr_amd import capi
    from rtlsdr_amd import build as hipbuild
    hipbuild.build()
    lib = capi.load()
    rng = np.random.default_rng(77)
    ref = oracle_lib.PowerReference()
    try:
        ref.lib.ref_frequency_range.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        for _ in range(250):
            lo = int(rng.integers(24_000_000, 1_600_000_000))
            width = int(rng.choice([rng.integers(20_000, 900_000), rng.integers(900_000, 3_000_000),
                                    rng.integers(3_000_000, 400_000_000)]))
            step = int(rng.choice([rng.integers(500, 20_000), rng.integers(20_000, 900_000),
                                   rng.integers(1_000_000, 2_500_000)]))
            crop = float(rng.choice([0.0, 0.0, 0.1, 0.25, 0.5]))
            boxcar = int(rng.integers(0, 2))
            arg = f"{lo}:{lo + width}:{step}"
            plan = capi.RtlpowerPlan()
            r = lib.rtlpower_frequency_range(lo, lo + width, step, crop, boxcar, C.byref(plan))
            if r != 0:
                continue  # rejected (the reference would exit or misbehave); covered by the fixed cases
            o = np.zeros(10, dtype=np.int32); rc = C.c_double()
            ref.lib.ref_frequency_range(arg.encode(), crop, boxcar, o.ctypes.data, C.byref(rc))
            assert plan.tune_count == o[0], (arg, crop, boxcar)
            if plan.tune_count == 0:
                continue  # an empty plan (giant bins wider than the range): nothing else is defined
            last = plan.tune_count - 1
            got = (plan.rate, plan.bin_e, plan.downsample, plan.downsample_passes, plan.buf_len,
                   lib.rtlpower_tune_freq(C.byref(plan), 0), lib.rtlpower_tune_freq(C.byref(plan), last), plan.crop)
            want = (o[2], o[3], o[4], o[5], o[6], o[7], o[8], rc.value)
            assert got == want, (arg, crop, boxcar)
    finally:
        ref.close()


def test_every_plan_of_the_planner_is_accepted():
    """What frequency_range() can plan, rtlpower_gpu_create() takes: bins from 2^1 to 2^21 (src/rtl_power.c:483-486),
    boxcar or fifth_order in front, any crop - rtlpower_cfg_validate (no GPU needed) must accept the configuration of
    every plan (rounds 1-3 answered -ENOTSUP above 2^14 bins)."""
    from rtlsdr_amd import capi
    from rtlsdr_amd import build as hipbuild
    hipbuild.build()
    lib = capi.load()
    rng = np.random.default_rng(78)
    seen = set()
    for k in range(600):
        lo = int(rng.integers(24_000_000, 1_600_000_000))
        width = int(rng.choice([rng.integers(20_000, 900_000), rng.integers(900_000, 3_000_000), rng.integers(3_000_000, 40_000_000)]))
        step = int(rng.choice([rng.integers(1, 50), rng.integers(50, 2000), rng.integers(2000, 900_000)]))
        crop = float(rng.choice([0.0, 0.1, 0.25]))
        boxcar = int(rng.integers(0, 2))
        plan = capi.RtlpowerPlan()
        if lib.rtlpower_frequency_range(lo, lo + width, step, crop, boxcar, C.byref(plan)) != 0 or plan.tune_count == 0:
            continue
        cfg = capi.RtlpowerCfg()
        lib.rtlpower_plan_cfg(C.byref(plan), int(rng.integers(0, 8)), boxcar, int(rng.choice([0, 9])), 0, C.byref(cfg))
        assert lib.rtlpower_cfg_validate(C.byref(cfg)) == 0, (lo, width, step, crop, boxcar, plan.bin_e, plan.downsample, plan.buf_len)
        seen.add(plan.bin_e)
    assert max(seen) == 21 and min(seen) <= 8, sorted(seen)
    bad = capi.RtlpowerCfg.default(bin_e=22, buf_len=1 << 24)
    assert lib.rtlpower_cfg_validate(C.byref(bad)) < 0
