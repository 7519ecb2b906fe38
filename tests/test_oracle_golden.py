"""The CPU oracle against the golden vectors the reference itself produced,
and (where oracle/_ref exists) against the live reference."""
import numpy as np
import pytest

import golden_util as gu
from cases import CASES, make_cfg
from rtlsdr_amd import capi, synth


@pytest.mark.parametrize("name", gu.fixture_names())
def test_oracle_matches_golden(oracle_lib, name):
    cfg, iq, want, want_state = gu.load(name)
    got, st = oracle_lib.run_stream(cfg, iq)
    assert got.shape == want.shape
    assert np.array_equal(got, want)  # bit-exact, -A std included (same libm)
    assert gu.state_dict(st) == gu.state_dict(want_state)


def test_manifest_hashes():
    import hashlib
    m = gu.manifest()
    for fn in gu.fixture_names():
        _, _, out, _ = gu.load(fn)
        assert hashlib.sha256(out.tobytes()).hexdigest() == m[fn]["out_sha256"]


def test_optimal_settings_known_answers(oracle_lib):
    import ctypes as C
    from rtlsdr_amd.capi import RtlfmCfg
    for row in gu.manifest()["_optimal_settings"]:
        cfg = RtlfmCfg.default(mode=row["mode"])
        cf, cr = C.c_uint32(), C.c_uint32()
        oracle_lib.oracle().orc_optimal_settings(
            C.byref(cfg), row["freq"], row["rate_in"], row["min_capture_rate"],
            row["use_fifth_order"], 0, C.byref(cf), C.byref(cr))
        assert (cfg.downsample, cfg.downsample_passes, cfg.output_scale, cf.value, cr.value) == (
            row["downsample"], row["downsample_passes"], row["output_scale"],
            row["capture_freq"], row["capture_rate"])


def test_deemph_a_known_answers(oracle_lib):
    # (rate_out, 75us, 50us)
    table = {170000: (13, 9), 240000: (19, 13), 150000: (12, 8), 32000: (3, 2),
             24000: (2, 2), 16000: (2, 1)}
    f = oracle_lib.oracle().orc_deemph_a
    for rate, (us, eu) in table.items():
        assert (f(rate, 75), f(rate, 50)) == (us, eu)


def test_fifth_order_block_boundary_quirk(oracle_lib):
    """x[N-1] of a block never reaches the next block's history (a6)."""
    lib = oracle_lib.oracle()
    rng = np.random.default_rng(5)
    x = rng.integers(-128, 129, size=64).astype(np.int16)
    inter = np.zeros(128, dtype=np.int16)
    inter[0::2] = x
    hist = np.zeros(6, dtype=np.int16)
    lib.orc_fifth_order(inter.ctypes.data, 128, hist.ctypes.data)
    assert list(hist) == list(x[57:63])  # x[N-7 .. N-2]; x[63] is gone


def test_arbitrary_upsample_ramp(oracle_lib):
    # SURVEY.md §8 a18 verified sample: ramp 0,100,... 128 -> 176
    lib = oracle_lib.oracle()
    a = (np.arange(128) * 100).astype(np.int16)
    b = np.zeros(176, dtype=np.int16)
    lib.orc_arbitrary_upsample(a.ctypes.data, b.ctypes.data, 128, 176)
    assert list(b[:4]) == [0, 72, 145, 218] and b[-1] == 12700


def test_polar_disc_lut_quirk(oracle_lib):
    lib = oracle_lib.oracle()
    # tiny positive angle, neither product zero: falls into the final else
    assert lib.orc_polar_disc_lut(1000, 1, 1000, 0) == 16384
    assert lib.orc_polar_disc_lut(-1000, -1, 1000, 0) in (0, -0)


@pytest.mark.parametrize("name,ov,sig", CASES[:12])
def test_oracle_matches_live_reference(oracle_lib, name, ov, sig):
    if not oracle_lib.have_reference():
        pytest.skip("oracle/_ref not built here")
    L, nb = 1024, 7
    cfg = make_cfg(ov, L)
    iq = synth.fm_iq_u8(1, L // 2 * nb, seed=99, **sig)[0]
    got, st = oracle_lib.run_stream(cfg, iq)
    ref = oracle_lib.Reference()
    want, rst = ref.run_stream(cfg, iq)
    ref.close()
    assert np.array_equal(got, want)
    assert gu.state_dict(st) == gu.state_dict(rst)


def _random_ref_cfg(rng):
    """A random configuration the reference itself can run (no division by zero in low_pass_real,
    passes within cic_9_tables, amplitudes that keep -A fast clear of its int32 wrap)."""
    from rtlsdr_amd import capi
    ov = {"mode": int(rng.choice([capi.MODE_FM] * 5 + [capi.MODE_AM, capi.MODE_USB, capi.MODE_LSB, capi.MODE_RAW]))}
    if rng.random() < 0.55:
        p_ = int(rng.integers(1, 9))
        ov.update(downsample=1 << p_, downsample_passes=p_, comp_fir_size=int(rng.choice([0, 9])))
    else:
        ov.update(downsample=int(rng.choice([1, 2, 3, 5, 6, 10, 16, 25, 42, 64, 84, 100, 255, 256, 334, 500])), downsample_passes=0)
    ov["custom_atan"] = int(rng.integers(0, 3))
    ov["offset_tuning"] = int(rng.random() < 0.25)
    ov["output_scale"] = int(rng.choice([1, 1, 2, 5]))
    ov["rate_out"] = int(rng.choice([8000, 16000, 24000, 48000, 170000]))
    if rng.random() < 0.4:
        ov.update(deemph=1, deemph_a=int(rng.choice([1, 2, 3, 9, 13, 19, 400])))
    if rng.random() < 0.25:
        ov["dc_block_audio"] = 1
    if rng.random() < 0.2:
        ov["dc_block_raw"] = 1
    if rng.random() < 0.2:
        ov["post_downsample"] = int(rng.choice([2, 3, 4]))
    if rng.random() < 0.2:
        ov["squelch_level"] = int(rng.choice([5, 50, 2000]))
    r = rng.random()
    if r < 0.25:
        ov.update(rate_out2=int(ov["rate_out"] * rng.choice([0.25, 0.4, 0.5, 0.9])), resampler=capi.RESAMPLE_LOW_PASS_REAL)
    elif r < 0.45:
        ov.update(rate_out2=int(ov["rate_out"] * rng.choice([0.5, 0.73, 1.38, 2.0])), resampler=capi.RESAMPLE_ARBITRARY)
    return ov


@pytest.mark.parametrize("seed", range(60))
def test_oracle_random_configurations_vs_live_reference(oracle_lib, seed):
    """The restatement against the reference's own full_demod() (compiled in place) on seeded
    random configurations: every sample and the complete carried state."""
    if not oracle_lib.have_reference():
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(31000 + seed)
    ov = _random_ref_cfg(rng)
    L = int(rng.choice([1024, 2048, 4096, 8192, 16384]))
    if ov["downsample_passes"]:
        while L % (2 << ov["downsample_passes"]) or (L // 2) >> ov["downsample_passes"] < 8:
            L *= 2
    nb = int(rng.integers(2, 6))
    cfg = make_cfg(ov, L)
    # low_pass_simple is only defined for "length multiple of step" (src/rtl_fm.c:740): past that
    # the reference sums stale samples beyond result_len.  Every other per-buffer stage works on
    # whatever count the buffer has.
    per = (L // 2) // ov["downsample"] if ov["downsample_passes"] == 0 else (L // 2) >> ov["downsample_passes"]
    uneven = ov["downsample_passes"] == 0 and (L // 2) % ov["downsample"]
    if ov.get("post_downsample", 1) > 1 and (uneven or per % ov["post_downsample"]):
        pytest.skip("post_downsample does not divide the buffer's output: outside low_pass_simple's domain")
    amp = 55.0
    if ov["custom_atan"] == 1 and ov["mode"] == 0:
        gain = ov["downsample"] if ov["downsample_passes"] == 0 else 1 << ov["downsample_passes"]
        amp = max(2.0, min(40.0, 500.0 / gain))
    iq = synth.fm_iq_u8(1, L // 2 * nb, seed=600 + seed, fs=1.024e6, dev_hz=20e3, amplitude=amp)[0]
    try:
        got, st = oracle_lib.run_stream(cfg, iq)
    except RuntimeError:
        pytest.skip("configuration rejected by the oracle")
    ref = oracle_lib.Reference()
    try:
        want, rst = ref.run_stream(cfg, iq)
    finally:
        ref.close()
    assert np.array_equal(got, want), (ov, L, nb)
    assert gu.state_dict(st) == gu.state_dict(rst), (ov, L, nb)


@pytest.mark.parametrize("passes,L", [(9, 1536), (9, 512 * 5), (9, 512 * 13), (10, 2560), (10, 3072), (10, 512 * 7), (10, 512 * 30), (9, 512 * 30)])
@pytest.mark.parametrize("kind", ["fm", "fmfir", "fmfast", "fmlutsq", "am", "usb", "raw"])
def test_oracle_vs_live_reference_on_buffers_the_passes_do_not_divide(oracle_lib, passes, L, kind):
    """`rtl_fm -W n -F 9` with nine or ten passes on buffers of 512 n bytes that 2^(passes + 1) does not divide
    (src/rtl_fm.c:1188-1191): the last passes see lengths that are not multiples of four elements.  The oracle's
    loops against the reference's own, compiled in place: every sample and the carried state.  (Buffers that come down
    to fewer than two elements make the reference read lowpassed[-1]: not compared here.)"""
    if not oracle_lib.have_reference():
        pytest.skip("oracle/_ref not built here")
    assert (L >> passes) >= 2
    ov = dict(downsample=1 << passes, downsample_passes=passes, comp_fir_size=9 if "fir" in kind else 0, rate_out=2000)
    if "fast" in kind: ov["custom_atan"] = 1
    if "lut" in kind: ov["custom_atan"] = 2
    if "sq" in kind: ov["squelch_level"] = 2000
    if kind == "am": ov.update(mode=capi.MODE_AM, output_scale=2)
    if kind == "usb": ov.update(mode=capi.MODE_USB, output_scale=3)
    if kind == "raw": ov["mode"] = capi.MODE_RAW
    cfg = make_cfg(ov, L)
    nb = 9
    iq = synth.fm_iq_u8(1, L // 2 * nb, seed=900 + passes + L, fs=1.024e6, dev_hz=300.0, amplitude=20.0 if "fast" in kind else 50.0)[0]
    got, st = oracle_lib.run_stream(cfg, iq)
    ref = oracle_lib.Reference()
    try:
        want, rst = ref.run_stream(cfg, iq)
    finally:
        ref.close()
    assert np.array_equal(got, want), (kind, passes, L)
    assert gu.state_dict(st) == gu.state_dict(rst), (kind, passes, L)


def test_batch_threads_equal_serial(oracle_lib):
    ov, sig = CASES[1][1], CASES[1][2]
    cfg = make_cfg(ov, 2048)
    iq = synth.fm_iq_u8(5, 1024 * 3, **sig)
    o1, n1, _ = oracle_lib.run_batch(cfg, iq, nthreads=1)
    o4, n4, _ = oracle_lib.run_batch(cfg, iq, nthreads=4)
    assert np.array_equal(o1, o4) and np.array_equal(n1, n4)
    for s in range(5):
        o, _ = oracle_lib.run_stream(cfg, iq[s])
        assert np.array_equal(o, o1[s, :n1[s]])


def test_rotate_90_u8_oracle_vs_reference_and_known_answer(oracle_lib):
    """rotate_90 on raw bytes (src/rtl_fm.c:437-447, dead code in the reference): the
    restatement against the reference's own function (when built here) and a known answer."""
    po = oracle_lib
    lib = po.oracle()
    rng = np.random.default_rng(90)
    for n in (8, 16, 512, 16384):
        x = rng.integers(0, 256, n, dtype=np.uint8)
        got = x.copy()
        lib.orc_rotate_90_u8(got.ctypes.data, n)
        # sample n times (+j)^n with NEG_U8(v) = 255 - v
        want = x.copy().reshape(-1, 8)
        src = x.reshape(-1, 8)
        want[:, 2] = 255 - src[:, 3]; want[:, 3] = src[:, 2]
        want[:, 4] = 255 - src[:, 4]; want[:, 5] = 255 - src[:, 5]
        want[:, 6] = src[:, 7]; want[:, 7] = 255 - src[:, 6]
        assert np.array_equal(got, want.ravel())
        if po.have_reference():
            ref = po.Reference()
            r = x.copy()
            ref.lib.ref_rotate_90_u8(r.ctypes.data, n)
            assert np.array_equal(r, got)
            ref.close()
