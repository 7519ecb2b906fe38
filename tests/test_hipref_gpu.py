"""The reference's OWN rtl_fm / rtl_power, with the hot path handed to the HIP layer.

oracle/make_hipref.py applies INTEGRATION.md's edits (callback body -> rtlfm_gpu_push, full_demod(d) ->
rtlfm_gpu_run + _fetch, set-up after optimal_settings(); scanner()'s body behind rtlsdr_read_sync ->
rtlpower_gpu_scan) to a temporary copy of /root/reference/src/rtl_fm.c / rtl_power.c and links it
against librtlfm_hip.so + the file-backed device layer: oracle/_ref/rtl_fm_hipref, rtl_power_hipref
(built in the build container, they travel like the other _ref binaries).  Everything outside the hot
path - getopt, optimal_settings, the -M wbfm preset, controller / dongle / demod / output threads,
fwrite - is the reference's code, unchanged.

north_star: "rtl_fm and rtl_power keep their CLI and rtlsdr_read_async callback plumbing but hand the
uint8 IQ ring buffer to a thin C-ABI HIP layer" - these tests execute that sentence.
"""
import ctypes as C
import os
import signal
import subprocess
import time

import numpy as np
import pytest

from rtlsdr_amd import capi, synth
from rtlsdr_amd.capi import ATAN_FAST, RESAMPLE_LOW_PASS_REAL, RtlfmCfg

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FM = os.path.join(ROOT, "oracle", "_ref", "rtl_fm_hipref")
POWER = os.path.join(ROOT, "oracle", "_ref", "rtl_power_hipref")


def _run_rtl_fm(args, src, out, want_bytes, env_extra=None):
    """The tool never exits on end of input (its main() polls do_exit, src/rtl_fm.c:2010-2012):
    SIGINT once everything expected is on disk or the output has stopped growing."""
    env = dict(os.environ, RTLSDR_FILE=str(src), RTLFM_HIPREF_LOSSLESS="1")
    env.update(env_extra or {})
    p = subprocess.Popen([FM] + args + [str(out)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    try:
        last, still = -1, 0
        for _ in range(600):  # at most 60 s (the first HIP call of a fresh box takes a while)
            time.sleep(0.1)
            sz = out.stat().st_size if out.exists() else 0
            still = still + 1 if sz == last and sz > 0 else 0
            last = sz
            if sz >= want_bytes or still >= 30 or p.poll() is not None:
                break
        time.sleep(0.2)
        if p.poll() is None:
            p.send_signal(signal.SIGINT)
        p.wait(timeout=20)
    finally:
        if p.poll() is None:
            p.kill()
    return p.stderr.read().decode(errors="replace")


CASES = [
    # BASELINE config 0: rtl_fm, 2.4 MS/s u8 IQ from a file, boxcar /10 + -A fast
    ("config0", ["-f", "100M", "-M", "fm", "-s", "240k", "-m", "2.2M", "-A", "fast"],
     dict(downsample=10, custom_atan=ATAN_FAST, rate_out=240000), dict(fs=2.4e6, dev_hz=75e3, amplitude=40.0)),
    # C2's command line: -M fm -s 150k -m 1.3M -F 0 -> 4 fifth_order passes at 2.4 MS/s
    ("c2", ["-f", "100M", "-M", "fm", "-s", "150k", "-m", "1.3M", "-F", "0"],
     dict(downsample=16, downsample_passes=4, rate_out=150000), dict(fs=2.4e6, dev_hz=75e3)),
    # C3's chain as far as the reference's CLI reaches it: -s 16k -F 9 -E deemp (live resampler off)
    ("c3", ["-f", "100M", "-M", "fm", "-s", "16k", "-F", "9", "-E", "deemp"],
     dict(downsample=64, downsample_passes=6, comp_fir_size=9, deemph=1, deemph_a=2, rate_out=16000),
     dict(fs=1.024e6, dev_hz=2.5e3)),
    # the -M wbfm preset: 170k, boxcar /6, -A fast, deemph, low_pass_real -> 32k
    ("wbfm", ["-f", "100M", "-M", "wbfm"],
     dict(downsample=6, custom_atan=ATAN_FAST, deemph=1, deemph_a=13, rate_out=170000, rate_out2=32000,
          resampler=RESAMPLE_LOW_PASS_REAL), dict(fs=1.02e6, dev_hz=75e3, amplitude=30.0)),
]


@pytest.mark.skipif(not os.path.exists(FM), reason="oracle/_ref/rtl_fm_hipref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("name,args,ov,sig", CASES, ids=[c[0] for c in CASES])
def test_reference_rtl_fm_on_the_hip_layer(oracle_lib, tmp_path, name, args, ov, sig):
    L, nb = 16384, 48  # the reference's default buffer (dongle_init, src/rtl_fm.c:1605)
    iq = synth.fm_iq_u8(1, L // 2 * nb, seed=61, **sig)[0]
    src, out = tmp_path / "cap.bin", tmp_path / "pcm.raw"
    iq.tofile(src)
    cfg = RtlfmCfg.default(block_len=L, **ov)
    want, _ = oracle_lib.run_stream(cfg, iq)
    err = _run_rtl_fm(args, src, out, 2 * want.size)
    assert "demodulating on the HIP layer" in err, err[-1500:]
    got = np.fromfile(out, dtype=np.int16)
    assert got.size == want.size, (name, got.size, want.size, err[-1500:])
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    # integer chains bit-exact; -A std within the stated 1 LSB on <= 1e-4 of the samples
    if ov.get("custom_atan", 0) == ATAN_FAST:
        assert d.max() == 0, (name, int((d != 0).sum()))
    else:
        assert d.max() <= 1 and (d != 0).mean() <= 1e-4, (name, int(d.max()), int((d != 0).sum()))


@pytest.mark.skipif(not os.path.exists(FM), reason="oracle/_ref/rtl_fm_hipref not built")
def test_reference_rtl_fm_on_the_hip_layer_live_mode_drops_not_blocks(oracle_lib, tmp_path):
    """Without RTLFM_HIPREF_LOSSLESS the patched tool keeps the reference's lossy hand-off: fed faster
    than real time it drops buffers, never blocks the callback and never writes more than the input holds."""
    L, nb = 16384, 32
    iq = synth.fm_iq_u8(1, L // 2 * nb, seed=62, fs=2.4e6, dev_hz=75e3)[0]
    src, out = tmp_path / "cap.bin", tmp_path / "pcm.raw"
    iq.tofile(src)
    err = _run_rtl_fm(["-f", "100M", "-M", "fm", "-s", "150k", "-m", "1.3M", "-F", "0"], src, out, 1 << 40,
                      env_extra=dict(RTLFM_HIPREF_LOSSLESS="0"))
    assert "demodulating on the HIP layer" in err, err[-1500:]
    got = np.fromfile(out, dtype=np.int16) if out.exists() else np.zeros(0, np.int16)
    assert got.size <= nb * (L // 2 // 16)
    assert got.size % (L // 2 // 16) == 0  # whole buffers


@pytest.mark.skipif(not os.path.exists(POWER), reason="oracle/_ref/rtl_power_hipref not built")
@pytest.mark.parametrize("window", ["hamming", "blackman-harris"])
def test_reference_rtl_power_on_the_hip_layer(oracle_lib, tmp_path, window):
    """rtl_power -f 100M:102.048M:125 (BASELINE config 4's command line, one hop, 16384 bins) for one
    interval.  The number of passes depends on the wall clock; with a capture of exactly one read,
    looped, every pass integrates the same frame (the first retune() drops 4096 bytes,
    src/rtl_power.c:542-552, so the frame is the capture rotated by 4096), avg[] and samples grow in
    proportion and the dB line does not depend on the count."""
    lib = capi.load()
    plan = capi.RtlpowerPlan()
    assert lib.rtlpower_frequency_range(100000000, 102048000, 125, 0.0, 1, C.byref(plan)) == 0
    assert plan.tune_count == 1
    wid = {"hamming": 1, "blackman-harris": 3}[window]
    cfg = capi.RtlpowerCfg()
    lib.rtlpower_plan_cfg(C.byref(plan), wid, 1, 0, 0, C.byref(cfg))
    L = plan.buf_len
    iq = synth.fm_iq_u8(1, L // 2, fs=2.048e6, dev_hz=40e3, seed=910)[0]
    src, out = tmp_path / "cap.bin", tmp_path / "p.csv"
    iq.tofile(src)
    env = dict(os.environ, RTLSDR_FILE=str(src), RTLSDR_FILE_LOOP="1")
    r = subprocess.run([POWER, "-f", "100M:102.048M:125", "-w", window, "-i", "1", "-1", str(out)], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "scanning on the HIP layer" in r.stderr
    line = out.read_text().splitlines()[0].split(", ")
    assert [int(x) for x in line[2:4]] == [100000000, 102048000]
    passes = int(line[5])
    assert passes >= 1
    db = np.array([float(x) for x in line[6:]])
    frame = np.roll(iq, -4096).reshape(1, -1)
    avg, n = oracle_lib.power_scan_batch(cfg, frame)
    buf = C.create_string_buffer(1 << 20)
    a = (avg[0] * passes).copy()
    assert lib.rtlpower_csv_dbm(C.byref(plan), 0, a.ctypes.data, int(n[0]) * passes, buf, len(buf)) > 0
    want = np.array([float(x) for x in buf.value.decode().strip().split(", ")[4:]])
    assert db.shape == want.shape
    assert np.array_equal(db, want), (passes, float(np.abs(db - want).max()))
