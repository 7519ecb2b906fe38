"""rtl_power rows p1-p5 through the C ABI on a real MI355X: int64 accumulators
and sample counts bit-exact against the reference's golden output and the oracle."""
import os

import numpy as np
import pytest

from rtlsdr_amd import synth
from rtlsdr_amd.capi import RtlpowerCfg
from test_power_oracle import load_power, power_fixtures

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def gpu_scan(cfg, iq2d, split=None, options=None):
    from rtlsdr_amd.power import GpuPower
    ns = iq2d.shape[0]
    L = int(cfg.buf_len)
    with GpuPower(cfg, ns, 0) as g:
        for k, v in (options or {}).items():
            g.set_option(k, v)
        d = torch.from_numpy(np.ascontiguousarray(iq2d)).cuda()
        if split:
            nr = iq2d.shape[1] // L
            a = d[:, :split * L].contiguous(); b = d[:, split * L:].contiguous()
            g.scan_torch(a); g.scan_torch(b)
        else:
            g.scan_torch(d)
        g.sync()
        res = [g.fetch(s) for s in range(ns)]
    return res


@pytest.mark.parametrize("name", power_fixtures())
def test_power_golden_fixture(name):
    cfg, iq, want, n = load_power(name)
    res = gpu_scan(cfg, np.stack([iq, iq]))
    for avg, samples in res:
        assert samples == n
        assert np.array_equal(avg, want), name


@pytest.mark.parametrize("kw", [
    dict(bin_e=14, window=1, buf_len=32768),                       # BASELINE config 4
    dict(bin_e=14, window=3, buf_len=32768, peak_hold=1),
    dict(bin_e=11, window=2, buf_len=16384),
    dict(bin_e=5, window=7, buf_len=16384),
    dict(bin_e=1, window=0, buf_len=16384),
    dict(bin_e=9, window=1, downsample=3, boxcar=1, buf_len=16384),   # buf_len/ds not a multiple of 2N
    dict(bin_e=10, window=4, downsample=8, downsample_passes=3, boxcar=0, comp_fir_size=9, buf_len=16384),
    dict(bin_e=13, window=5, downsample=2, downsample_passes=1, boxcar=0, buf_len=32768),
    dict(bin_e=0, buf_len=16384),
    # beyond one workgroup's LDS: the transform over the work buffer in HBM (power_kernels.h, k_power_fft_*)
    dict(bin_e=14, window=1, buf_len=65536),                          # two frames of 16384 points per read
    dict(bin_e=10, window=6, buf_len=262144, peak_hold=1),            # 64 frames per read
    dict(bin_e=15, window=1, buf_len=65536),
    dict(bin_e=16, window=3, downsample=2, boxcar=1, buf_len=262144),
    dict(bin_e=15, window=2, downsample=2, downsample_passes=1, boxcar=0, comp_fir_size=9, buf_len=131072),
    dict(bin_e=18, window=5, buf_len=524288),
])
def test_power_batched_vs_oracle(oracle_lib, kw):
    cfg = RtlpowerCfg.default(**kw)
    L, nr, ns = int(cfg.buf_len), 5, 12
    if L > 65536:
        nr, ns = 3, 4
    iq = np.concatenate([synth.fm_iq_u8(ns // 2, L // 2 * nr, fs=2.048e6, dev_hz=30e3, seed=77),
                         synth.random_u8(ns // 2, L * nr, seed=78)])
    want, wn = oracle_lib.power_scan_batch(cfg, iq, nthreads=4)
    for split in (None, 2):
        res = gpu_scan(cfg, iq, split=split)
        for s in range(ns):
            assert res[s][1] == wn[s]
            assert np.array_equal(res[s][0], want[s]), (kw, s, split)


def _skip_only_outside_scanner_domain(cfg, err):
    """rtlpower_gpu_create may only reject what scanner() itself cannot run as a function of its input
    (include/rtlpower_hip.h, rtlpower_cfg_validate): a trailing FFT frame that reaches past the read - there the
    reference transforms what its static fft_buf still holds from the read before (src/rtl_power.c:695) -, a
    partial trailing frame behind fifth_order passes, or reads too short for the passes' ease-in.  Its own
    planner (frequency_range, :501-504: buf_len = 2N * downsample, at least 16384) produces none of these.
    Anything else the library refuses fails the test."""
    two_n = 2 << int(cfg.bin_e)
    ds = max(1, int(cfg.downsample))
    per = int(cfg.buf_len) // ds
    frames = (per + two_n - 1) // two_n
    passes = int(cfg.downsample_passes) if not cfg.boxcar else 0
    outside = frames * two_n > int(cfg.buf_len) or (passes and per % two_n) or (passes and (int(cfg.buf_len) >> passes) < 48)
    if cfg.bin_e == 0:
        outside = False
    if outside:
        pytest.skip(f"outside scanner()'s own domain: {err}")
    raise AssertionError(f"the library rejects a configuration the reference's scanner() runs: {err} "
                         f"(bin_e={cfg.bin_e} buf_len={cfg.buf_len} ds={cfg.downsample} passes={cfg.downsample_passes} boxcar={cfg.boxcar})")


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTLFM_SWEEP_POWER", "40"))))
def test_power_random_configurations(oracle_lib, seed):
    """Seeded random rtl_power configurations (bin size, window, boxcar / fifth_order + FIR
    decimation, peak hold, read size, streams, reads split over launches) against the oracle."""
    from rtlsdr_amd import capi
    from rtlsdr_amd.power import GpuPower
    rng = np.random.default_rng(5000 + seed)
    kw = dict(bin_e=int(rng.integers(0, 15)), window=int(rng.integers(0, 8)), peak_hold=int(rng.random() < 0.25))
    kw["buf_len"] = int(rng.choice([16384, 16384, 32768, 65536]))
    r = rng.random()
    if r < 0.3:
        kw.update(downsample=int(rng.choice([2, 3, 5, 7, 16])), boxcar=1)
    elif r < 0.6:
        p_ = int(rng.integers(1, 5))
        kw.update(downsample=1 << p_, downsample_passes=p_, boxcar=0, comp_fir_size=int(rng.choice([0, 9])))
    cfg = RtlpowerCfg.default(**kw)
    ns = int(rng.choice([1, 3, 10]))
    try:
        GpuPower(cfg, ns, 0).close()
    except capi.RtlfmError as e:
        _skip_only_outside_scanner_domain(cfg, e)
    L, nr = int(cfg.buf_len), int(rng.integers(2, 5))
    iq = np.concatenate([synth.fm_iq_u8(ns, L // 2 * nr, fs=2.048e6, dev_hz=40e3, seed=100 + seed),
                         synth.random_u8(1, L * nr, seed=200 + seed)])
    want, wn = oracle_lib.power_scan_batch(cfg, iq, nthreads=2)
    # scan_frames = 0: the general in-LDS kernel also where the one written for "several frames per read" applies
    for split, frames in ((None, 1), (1, 1), (None, 0)):
        res = gpu_scan(cfg, iq, split=split, options=dict(scan_frames=frames))
        for s in range(ns + 1):
            assert res[s][1] == wn[s], (kw, s, split, frames)
            assert np.array_equal(res[s][0], want[s]), (kw, s, split, frames)



def _decimated_cases():
    """What frequency_range() plans for a range below 1 MHz (src/rtl_power.c:466-504): buf_len = 2 N ds, at least 16384."""
    cases = []
    for passes in (1, 2, 3, 4, 5, 6):
        for bin_e in (3, 7, 9, 10, 11, 12, 13):
            L = max(16384, (2 << bin_e) << passes)
            per = (L >> passes) // 2
            if per < 512 or per > 8192:
                continue
            for fir in (0, 9):
                cases.append(dict(bin_e=bin_e, window=(bin_e + passes) % 8, downsample=1 << passes, downsample_passes=passes, boxcar=0,
                                  comp_fir_size=fir, buf_len=L, peak_hold=int((bin_e + passes) % 5 == 0)))
    for ds in (2, 4, 7, 8, 14, 16):
        for bin_e in (5, 9, 10, 12):
            L = max(16384, (2 << bin_e) * ds)
            cases.append(dict(bin_e=bin_e, window=(bin_e + ds) % 8, downsample=ds, boxcar=1, buf_len=L))
    return cases


@pytest.mark.parametrize("kw", _decimated_cases(), ids=lambda kw: "-".join(f"{k[:3]}{v}" for k, v in kw.items() if k in ("bin_e", "downsample", "boxcar", "comp_fir_size", "buf_len")))
def test_power_decimated_scans(oracle_lib, kw):
    """A scan of a range below 1 MHz decimates in front of the FFT (downsample_iq x passes + generic_fir with -F, the boxcar
    without: src/rtl_power.c:671-691).  Where the decimated read is whole frames of 512 ... 8192 points the library takes
    k_power_downsample_iq / k_power_boxcar + k_power_scan_frames<13, true> (last_kernel == 4); the general kernels (option
    dec_fast = 0) and the oracle must give the same integers: FM signal and full-scale bytes, read counts that do and do
    not fill the scan kernel's groups of reads, reads split over launches and over workgroups."""
    from rtlsdr_amd.power import GpuPower
    cfg = RtlpowerCfg.default(**kw)
    L = int(cfg.buf_len)
    per = (L // int(cfg.downsample)) // 2
    fast_expected = (512 <= per <= 8192 and per & (per - 1) == 0 and (L // 2) % int(cfg.downsample) == 0 and (1 << cfg.bin_e) <= per)
    for nr, ns in ((5, 3), (16, 2)):
        iq = np.concatenate([synth.fm_iq_u8(ns, L // 2 * nr, fs=2.048e6, dev_hz=30e3, seed=300 + cfg.bin_e),
                             synth.random_u8(1, L * nr, seed=400 + cfg.bin_e)])
        want, wn = oracle_lib.power_scan_batch(cfg, iq, nthreads=4)
        for opts, split in ((dict(), None), (dict(), 2), (dict(groups=3), None), (dict(dec_fast=0), None)):
            with GpuPower(cfg, ns + 1, 0) as g:
                for k, v in opts.items():
                    g.set_option(k, v)
                d = torch.from_numpy(iq).cuda()
                if split:
                    g.scan_torch(d[:, :split * L].contiguous()); g.scan_torch(d[:, split * L:].contiguous())
                else:
                    g.scan_torch(d)
                g.sync()
                assert (g.last_kernel == 4) == (fast_expected and opts.get("dec_fast", 1) == 1), (kw, opts, g.last_kernel)
                for s in range(ns + 1):
                    avg, samples = g.fetch(s)
                    assert samples == wn[s], (kw, opts, split, s)
                    assert np.array_equal(avg, want[s]), (kw, opts, split, s, nr)

@pytest.mark.parametrize("bin_e", list(range(3, 14)))
@pytest.mark.parametrize("L", [16384, 32768])
def test_power_several_frames_per_read(oracle_lib, bin_e, L):
    """rtl_power's everyday shape - the planner never reads less than 16384 bytes (src/rtl_power.c:501-504), so a scan
    below 8192 bins hands over reads that hold several frames - on k_power_scan_frames and on the general kernel, against
    the oracle: every bin size, both read sizes, peak hold, full-scale bytes, few and many streams (reads split over
    workgroups)."""
    if (2 << bin_e) >= L:
        pytest.skip("one frame per read: k_power_scan_big's case")
    cfg = RtlpowerCfg.default(bin_e=bin_e, window=(bin_e % 7) + 1, buf_len=L, peak_hold=int(bin_e % 4 == 1))
    nr, ns = 5, 3
    iq = np.concatenate([synth.fm_iq_u8(ns - 1, L // 2 * nr, fs=2.048e6, dev_hz=40e3, seed=900 + bin_e),
                         synth.random_u8(1, L * nr, seed=950 + bin_e)])
    want, wn = oracle_lib.power_scan_batch(cfg, iq, nthreads=2)
    for split, frames in ((None, 1), (2, 1), (None, 0)):
        res = gpu_scan(cfg, iq, split=split, options=dict(scan_frames=frames))
        for s in range(ns):
            assert res[s][1] == wn[s], (bin_e, L, s, split, frames)
            assert np.array_equal(res[s][0], want[s]), (bin_e, L, s, split, frames)


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTLFM_SWEEP_POWER_BIG", "10"))))
def test_power_fine_bins_random(oracle_lib, seed):
    """frequency_range()'s fine-bin plans, 2^15 .. 2^18 bins (src/rtl_power.c:483-486: up to 2^21): windows,
    boxcar / fifth_order decimation in front, peak hold, reads split over launches, against the oracle."""
    from rtlsdr_amd import capi
    from rtlsdr_amd.power import GpuPower
    rng = np.random.default_rng(15000 + seed)
    bin_e = int(rng.integers(15, 19))
    kw = dict(bin_e=bin_e, window=int(rng.integers(0, 8)), peak_hold=int(rng.random() < 0.25))
    ds = 1
    r = rng.random()
    if r < 0.25:
        ds = int(rng.choice([2, 3]))
        kw.update(downsample=ds, boxcar=1)
    elif r < 0.5:
        ds = 2
        kw.update(downsample=2, downsample_passes=1, boxcar=0, comp_fir_size=int(rng.choice([0, 9])))
    kw["buf_len"] = (2 << bin_e) * ds  # frequency_range(): buf_len = 2 * 2^bin_e * downsample (:501)
    cfg = RtlpowerCfg.default(**kw)
    ns = int(rng.choice([1, 2]))
    try:
        GpuPower(cfg, ns, 0).close()
    except capi.RtlfmError as e:
        _skip_only_outside_scanner_domain(cfg, e)
    L, nr = int(cfg.buf_len), int(rng.integers(2, 4))
    iq = np.concatenate([synth.fm_iq_u8(ns, L // 2 * nr, fs=2.048e6, dev_hz=40e3, seed=300 + seed),
                         synth.random_u8(1, L * nr, seed=400 + seed)])
    want, wn = oracle_lib.power_scan_batch(cfg, iq, nthreads=3)
    # staged_fast = 0: the general kernels also where the ones written for "one undecimated frame per read" apply
    for split, fast in ((None, 1), (1, 1), (None, 0)):
        res = gpu_scan(cfg, iq, split=split, options=dict(staged_fast=fast))
        for s in range(ns + 1):
            assert res[s][1] == wn[s], (kw, s, split, fast)
            assert np.array_equal(res[s][0], want[s]), (kw, s, split, fast)


@pytest.mark.parametrize("bin_e", [15, 16, 17, 18, 19, 20, 21])
def test_power_one_frame_per_read_every_size(oracle_lib, bin_e):
    """rtl_power's own fine-bin shape (an undecimated read = one frame of 2^bin_e points, src/rtl_power.c:483-501) on the
    kernels written for it - averages in slices, the bytes comb by comb, k_power_scan_big<14, true>, the last pass over
    HBM accumulating - and on the general ones, against the oracle; a stream stride that is not a multiple of 16 bytes
    takes the general kernels by itself."""
    cfg = RtlpowerCfg.default(bin_e=bin_e, window=int(bin_e % 7) + 1, buf_len=2 << bin_e, peak_hold=int(bin_e == 16))
    L, nr, ns = int(cfg.buf_len), 3, 2
    iq = np.concatenate([synth.fm_iq_u8(1, L // 2 * nr, fs=2.048e6, dev_hz=40e3, seed=700 + bin_e),
                         synth.random_u8(1, L * nr, seed=800 + bin_e)])
    want, wn = oracle_lib.power_scan_batch(cfg, iq, nthreads=2)
    for fast in (1, 0):
        res = gpu_scan(cfg, iq, split=1 if fast else None, options=dict(staged_fast=fast))
        for s in range(ns):
            assert res[s][1] == wn[s], (bin_e, s, fast)
            assert np.array_equal(res[s][0], want[s]), (bin_e, s, fast)
    # the batches of a scan as a two-stream pipeline (round 5): one read per batch - three batches through the two sets of work
    # buffers -, and the same on one stream
    for opts in (dict(staged_batch=1, staged_pipe=2), dict(staged_batch=1, staged_pipe=0), dict(staged_batch=2, staged_pipe=2), dict(staged_pipe=2),
                 dict(staged_batch=1)):
        for split in (None, 2):
            res = gpu_scan(cfg, iq, split=split, options=opts)
            for s in range(ns):
                assert res[s][1] == wn[s], (bin_e, s, opts, split)
                assert np.array_equal(res[s][0], want[s]), (bin_e, s, opts, split)
    if bin_e <= 17:
        from rtlsdr_amd.power import GpuPower
        wide = torch.zeros((ns, L * nr + 8), dtype=torch.uint8, device="cuda")
        wide[:, :L * nr] = torch.from_numpy(iq).cuda()
        with GpuPower(cfg, ns, 0) as g:
            g.scan_device(wide.data_ptr(), wide.stride(0), nr)
            g.sync()
            for s in range(ns):
                avg, n = g.fetch(s)
                assert n == wn[s] and np.array_equal(avg, want[s]), (bin_e, s, "stride")


def test_power_clock_probe():
    """rtlpower_gpu_clock_probe: the large-FFT kernel's workgroups stamp the shader clock they ran at."""
    from rtlsdr_amd.power import GpuPower
    cfg = RtlpowerCfg.default(bin_e=14, window=1, buf_len=32768)
    iq = torch.from_numpy(synth.random_u8(64, 32768 * 4, seed=3)).cuda()
    with GpuPower(cfg, 64, 0) as g:
        assert g.clock_read() is None
        g.clock_probe(True)
        g.scan_torch(iq)
        mhz, span = g.clock_read()
        assert 500.0 < mhz < 3500.0 and 0 < span < 1000.0, (mhz, span)
        g.clock_probe(False)


def test_power_host_scanner_and_clear(oracle_lib):
    """rtlsdr_read_sync-shaped entry point (one tuning state, one host buffer) + csv_dbm's reset."""
    from rtlsdr_amd.power import GpuPower
    cfg = RtlpowerCfg.default(bin_e=10, window=1, buf_len=16384)
    iq = synth.fm_iq_u8(3, 8192 * 2, fs=2.048e6, seed=5)
    want, wn = oracle_lib.power_scan_batch(cfg, iq, nthreads=1)
    with GpuPower(cfg, 3, 0) as g:
        for r in range(2):
            for s in range(3):
                g.scanner(iq[s, r * 16384:(r + 1) * 16384], s)
        for s in range(3):
            avg, n = g.fetch(s)
            assert n == wn[s] and np.array_equal(avg, want[s])
        g.clear()
        avg, n = g.fetch(1)
        assert n == 0 and not avg.any()


def test_power_window_coefs_match_oracle(oracle_lib):
    from rtlsdr_amd.power import window_coefs
    for w in range(8):
        for n in (2, 16, 1024, 16384):
            assert np.array_equal(window_coefs(w, n), oracle_lib.power_window_coefs(w, n)), (w, n)


def test_power_c4_full_size(oracle_lib):
    """configs[3]: 1024 streams x 64 reads x 32768 B, 16k-bin FFT + integrate (2 GiB)."""
    from rtlsdr_amd.power import GpuPower
    S, NR, L = 1024, 64, 32768
    cfg = RtlpowerCfg.default(bin_e=14, window=1, buf_len=L)
    iq = synth.fm_iq_u8_torch(S, NR * L // 2, torch.device("cuda", 0), fs=2.048e6, dev_hz=50e3)
    with GpuPower(cfg, S, 0) as g:
        g.scan_torch(iq); g.sync()
        a0, n0 = g.fetch(0); a1, n1 = g.fetch(S - 1)
        # linearity of the integration: scanning the same reads again doubles every bin
        g.scan_torch(iq); g.sync()
        b0, m0 = g.fetch(0)
    assert n0 == NR and m0 == 2 * NR and np.array_equal(b0, 2 * a0)
    for s, (a, n) in ((0, (a0, n0)), (S - 1, (a1, n1))):
        want, wn = oracle_lib.power_scan_batch(cfg, iq[s:s + 1].cpu().numpy(), nthreads=1)
        assert n == wn[0] and np.array_equal(a, want[0])
