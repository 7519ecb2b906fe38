"""rtl_fm_hip (the rtl_fm-shaped CLI over librtlsdr_file + librtlfm_hip) against the
oracle: the PCM file it writes must be what rtl_fm's DSP produces for the same IQ file."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rtlsdr_amd import build as hipbuild
from rtlsdr_amd import synth
from rtlsdr_amd.capi import (ATAN_FAST, ATAN_STD, MODE_AM, MODE_FM, MODE_USB, RESAMPLE_LOW_PASS_REAL, RtlfmCfg)

pytestmark = pytest.mark.gpu

CASES = [
    # argv, planner inputs (rate_in, min_capture, fifth, cfg overrides), fs of the synthetic capture
    (["-M", "fm", "-s", "240k", "-m", "2.2M", "-A", "fast"], (240000, 2200000, 0, dict(custom_atan=ATAN_FAST, rate_out=240000)), 2.4e6),
    (["-M", "fm", "-s", "150k", "-m", "1.3M", "-F", "0"], (150000, 1300000, 1, dict(rate_out=150000)), 2.4e6),
    (["-s", "16k", "-F", "9", "-E", "deemp", "-W", "128"], (16000, 1000000, 1, dict(rate_out=16000, comp_fir_size=9, deemph=1, block_len=65536)), 1.024e6),
    (["-M", "wbfm"], (170000, 1000000, 0, dict(rate_out=170000, rate_out2=32000, custom_atan=ATAN_FAST, deemph=1, resampler=RESAMPLE_LOW_PASS_REAL)), 1.02e6),
    (["-M", "am", "-s", "24k", "-F", "9", "-E", "dc"], (24000, 1000000, 1, dict(mode=MODE_AM, rate_out=24000, comp_fir_size=9, dc_block_audio=1)), 1.536e6),
    # the reference's default plans: a boxcar that does not divide the 8192-sample buffer (/42, /84, /6, /334)
    # in front of the per-buffer stages
    (["-s", "24k", "-E", "dc"], (24000, 1000000, 0, dict(rate_out=24000, dc_block_audio=1)), 1.008e6),
    (["-M", "am", "-s", "12k", "-E", "dc"], (12000, 1000000, 0, dict(mode=MODE_AM, rate_out=12000, dc_block_audio=1)), 1.008e6),
    (["-M", "wbfm", "-E", "dc"], (170000, 1000000, 0, dict(rate_out=170000, rate_out2=32000, custom_atan=ATAN_FAST, deemph=1, resampler=RESAMPLE_LOW_PASS_REAL, dc_block_audio=1)), 1.02e6),
    (["-M", "usb", "-s", "3k"], (3000, 1000000, 0, dict(mode=MODE_USB, rate_out=3000)), 1.002e6),
    # -Z: the device layer reads straight into the pinned staging ring (rtlfm_gpu_acquire / _commit)
    (["-M", "fm", "-s", "150k", "-m", "1.3M", "-F", "0", "-Z"], (150000, 1300000, 1, dict(rate_out=150000)), 2.4e6),
    (["-M", "wbfm", "-Z"], (170000, 1000000, 0, dict(rate_out=170000, rate_out2=32000, custom_atan=ATAN_FAST, deemph=1, resampler=RESAMPLE_LOW_PASS_REAL)), 1.02e6),
    # -E rdc: dc_block_raw_filter in front of the chain, on the one-launch front end
    (["-M", "fm", "-s", "150k", "-m", "1.3M", "-F", "0", "-E", "rdc"], (150000, 1300000, 1, dict(rate_out=150000, dc_block_raw=1)), 2.4e6),
]


@pytest.mark.parametrize("argv,plan,fs", CASES)
def test_cli_output_matches_oracle(oracle_lib, tmp_path, argv, plan, fs):
    _, cli = hipbuild.build_host()
    rate_in, min_capture, fifth, ov = plan
    cfg = RtlfmCfg.default(**ov)
    cf, cr = C.c_uint32(), C.c_uint32()
    # -M wbfm tunes 16 kHz up (controller_thread_fn, src/rtl_fm.c:1455-1460)
    oracle_lib.oracle().orc_optimal_settings(C.byref(cfg), 100000000 + (16000 if "wbfm" in argv else 0), rate_in, min_capture,
                                             fifth, 0, C.byref(cf), C.byref(cr))
    if cfg.deemph:
        cfg.deemph_a = oracle_lib.oracle().orc_deemph_a(cfg.rate_out, 75)
    L = int(cfg.block_len)
    nb = 37
    amp = 30.0 if cfg.custom_atan == ATAN_FAST else 60.0
    iq = synth.fm_iq_u8(1, L // 2 * nb + 100, fs=fs, dev_hz=5e3, amplitude=amp, seed=404)[0]  # + a short tail
    src = tmp_path / "capture.bin"
    iq.tofile(src)
    out = tmp_path / "audio.raw"
    env = dict(os.environ, RTLSDR_FILE=str(src))
    r = subprocess.run([cli, "-f", "100M"] + argv + [str(out)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert f"Sampling at {cr.value} S/s" in r.stderr and f"Tuned to {cf.value} Hz" in r.stderr
    got = np.fromfile(out, dtype=np.int16)
    want, _ = oracle_lib.run_stream(cfg, iq[:L * nb])
    assert got.shape == want.shape
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    exact_needed = cfg.mode == MODE_FM and cfg.custom_atan != ATAN_STD
    assert d.max() <= (0 if exact_needed else 1) and (d != 0).mean() <= 1e-4, (argv, int(d.max()), int((d != 0).sum()))


@pytest.mark.parametrize("zero_copy", [True, False])
def test_cli_on_a_slow_rtl_tcp_source_loses_nothing(oracle_lib, tmp_path, zero_copy):
    """The dongle / demod hand-off under a source that dribbles (an rtl_tcp server that sends 3000 bytes at a time with
    pauses, RTLSDR_FILE=tcp://...): with -Z the device layer has a ring slot open nearly all the time and
    rtlfm_gpu_run() refuses to start while one is - no committed buffer may be left in the ring at the end of the
    stream, none demodulated twice.  The PCM must be the oracle's for every whole buffer that arrived."""
    import socket
    import threading
    import time
    _, cli = hipbuild.build_host()
    ov = dict(rate_out=150000)
    cfg = RtlfmCfg.default(**ov)
    cf, cr = C.c_uint32(), C.c_uint32()
    oracle_lib.oracle().orc_optimal_settings(C.byref(cfg), 100000000, 150000, 1300000, 1, 0, C.byref(cf), C.byref(cr))
    L, nb = int(cfg.block_len), 23
    iq = synth.fm_iq_u8(1, L // 2 * nb, fs=2.4e6, dev_hz=5e3, amplitude=60.0, seed=405)[0]
    payload = iq.tobytes()
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)
    port = srv.getsockname()[1]

    def serve():
        c, _ = srv.accept()
        c.sendall(b"RTL0" + (5).to_bytes(4, "big") + (29).to_bytes(4, "big"))
        for at in range(0, len(payload), 3000):
            c.sendall(payload[at:at + 3000])
            time.sleep(0.002 if (at // 3000) % 7 else 0.02)
        # the client's tuning commands sit unread in this socket: closing it now would answer with a reset and the
        # client's kernel would drop what it has not read yet - finish the sending side, then drain until the client leaves
        c.shutdown(socket.SHUT_WR)
        c.settimeout(20.0)
        try:
            while c.recv(4096):
                pass
        except OSError:
            pass
        c.close()
    t = threading.Thread(target=serve, daemon=True)
    t.start()
    out = tmp_path / "audio.raw"
    env = dict(os.environ, RTLSDR_FILE=f"tcp://127.0.0.1:{port}")
    argv = ["-M", "fm", "-s", "150k", "-m", "1.3M", "-F", "0"] + (["-Z"] if zero_copy else [])
    r = subprocess.run([cli, "-f", "100M"] + argv + [str(out)], env=env, capture_output=True, text=True, timeout=300)
    t.join(10)
    srv.close()
    assert r.returncode == 0, r.stderr[-1500:]
    assert f"{nb} buffers in" in r.stderr, r.stderr[-400:]
    got = np.fromfile(out, dtype=np.int16)
    want, _ = oracle_lib.run_stream(cfg, iq)
    assert got.shape == want.shape, (got.shape, want.shape, r.stderr[-300:])
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-4


@pytest.mark.parametrize("argv,passes", [
    (["-f", "100M:102.048M:125", "-w", "hamming", "-1"], 4),          # BASELINE config 4's command line
    (["-f", "88M:96M:10k", "-w", "blackman", "-c", "20%", "-1"], 3),  # 3 hops, cropped
    (["-f", "433M:434M:1k", "-F", "9", "-1"], 2),                      # one hop, fifth_order /2 + FIR9
    (["-f", "24M:34M:1M", "-1"], 2),                                   # giant bins: rms_power per hop
    (["-f", "100M:102M:50", "-w", "hamming", "-1"], 2),                # fine bins: 2^16 per hop, the transform over HBM
    (["-f", "144M:146.4M:10", "-c", "20%", "-1"], 2),                  # two hops of 2^18 bins, cropped
])
def test_rtl_power_cli_matches_oracle(oracle_lib, tmp_path, argv, passes):
    from rtlsdr_amd import capi
    lib = capi.load()
    cli = os.path.join(os.path.dirname(hipbuild.CLI_OUT), "rtl_power_hip")
    hipbuild.build_host()
    farg = argv[argv.index("-f") + 1]
    lo, hi, step = (int(float(t[:-1]) * {"k": 1e3, "M": 1e6}[t[-1]]) if t[-1] in "kM" else int(float(t)) for t in farg.split(":"))
    crop = 0.2 if "-c" in argv else 0.0
    boxcar = 0 if "-F" in argv else 1
    win = {"hamming": 1, "blackman": 2}.get(argv[argv.index("-w") + 1], 0) if "-w" in argv else 0
    plan = capi.RtlpowerPlan()
    assert lib.rtlpower_frequency_range(lo, hi, step, crop, boxcar, C.byref(plan)) == 0
    cfg = capi.RtlpowerCfg()
    lib.rtlpower_plan_cfg(C.byref(plan), win, boxcar, 9 if "-F" in argv else 0, 0, C.byref(cfg))
    L, T = plan.buf_len, plan.tune_count
    iq = synth.fm_iq_u8(1, L // 2 * T * passes, fs=2.048e6, dev_hz=40e3, seed=909)[0]
    src = tmp_path / "capture.bin"
    iq.tofile(src)
    out = tmp_path / "power.csv"
    env = dict(os.environ, RTLSDR_FILE=str(src), RTLPOWER_PASSES=str(passes))
    r = subprocess.run([cli] + argv + [str(out)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = out.read_text().splitlines()
    assert len(lines) == T
    # the file is read hop after hop, pass after pass
    reads = iq.reshape(passes, T, L)
    for i in range(T):
        avg, n = oracle_lib.power_scan_batch(cfg, np.ascontiguousarray(reads[:, i, :]).reshape(1, -1))
        buf = C.create_string_buffer(8 << 20)
        a = avg[0].copy()
        assert lib.rtlpower_csv_dbm(C.byref(plan), i, a.ctypes.data, int(n[0]), buf, len(buf)) > 0
        got = lines[i].split(", ", 2)[2]  # drop "date, time, "
        assert got + "\n" == buf.value.decode(), (i, got[:80], buf.value[:80])


def test_cli_squelch_holds_output_back_like_demod_thread_fn(oracle_lib, tmp_path):
    """-l: rtl_fm's demod thread forwards nothing while squelch_hits > conseq_squelch and clamps the
    counter (src/rtl_fm.c:1366-1370); squelch_hits starts at 11, so the tool is silent until the
    squelch first opens.  Capture: silence, signal, silence."""
    _, cli = hipbuild.build_host()
    L, nb = 16384, 60
    cfg = RtlfmCfg.default(rate_out=24000, squelch_level=40)
    cf, cr = C.c_uint32(), C.c_uint32()
    oracle_lib.oracle().orc_optimal_settings(C.byref(cfg), 100000000, 24000, 1000000, 0, 0, C.byref(cf), C.byref(cr))
    sig = synth.fm_iq_u8(1, L // 2 * nb, fs=1.008e6, dev_hz=2.5e3, amplitude=60.0, seed=77)[0].reshape(nb, L)
    quiet = synth.fm_iq_u8(1, L // 2 * nb, fs=1.008e6, dev_hz=2.5e3, amplitude=0.0, noise_lsb=1, seed=78)[0].reshape(nb, L)
    iq = np.concatenate([quiet[:8], sig[8:30], quiet[30:]]).ravel()
    src, out = tmp_path / "capture.bin", tmp_path / "audio.raw"
    iq.tofile(src)
    r = subprocess.run([cli, "-f", "100M", "-s", "24k", "-l", "40", "-t", "3", str(out)], env=dict(os.environ, RTLSDR_FILE=str(src)),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    got = np.fromfile(out, dtype=np.int16)
    # the same rule over the oracle's per-buffer outputs
    lib = oracle_lib.oracle()
    st = oracle_lib.new_states(1)[0]
    scratch = np.zeros(L, dtype=np.int16)
    want, held = [], 0
    for b in range(nb):
        n = lib.orc_block(C.byref(cfg), C.byref(st), np.ascontiguousarray(iq[b * L:(b + 1) * L]), L, scratch)
        if st.squelch_hits > 3:
            st.squelch_hits = 4
            held += 1
            continue
        want.append(scratch[:n].copy())
    want = np.concatenate(want)
    assert 20 < held < nb - 10 and f"{held} buffers held back" in r.stderr
    assert got.shape == want.shape and np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_cli_prints_levels_like_full_demod(oracle_lib, tmp_path):
    """-L n: the level line of full_demod() (src/rtl_fm.c:1217-1237) every n buffers — the first one
    right away (printLevelNo starts at 1), averages over n."""
    _, cli = hipbuild.build_host()
    L, nb, n_every = 16384, 9, 4
    cfg = RtlfmCfg.default(rate_out=150000)
    cf, cr = C.c_uint32(), C.c_uint32()
    oracle_lib.oracle().orc_optimal_settings(C.byref(cfg), 100000000, 150000, 1300000, 1, 0, C.byref(cf), C.byref(cr))
    iq = synth.fm_iq_u8(1, L // 2 * nb, fs=2.4e6, dev_hz=75e3, amplitude=55.0, seed=12)[0]
    src, out = tmp_path / "capture.bin", tmp_path / "audio.raw"
    iq.tofile(src)
    r = subprocess.run([cli, "-f", "100M", "-M", "fm", "-s", "150k", "-m", "1.3M", "-F", "0", "-L", str(n_every), str(out)],
                       env=dict(os.environ, RTLSDR_FILE=str(src)), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [ln for ln in r.stderr.splitlines() if " avg rms, " in ln]
    # the same bookkeeping over the oracle's rms() of the decimated IQ
    lib = oracle_lib.oracle()
    lib.orc_rms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.orc_rms.restype = C.c_int
    from rtlsdr_amd import capi
    raw = RtlfmCfg.from_buffer_copy(bytes(cfg)); raw.mode = capi.MODE_RAW
    st = oracle_lib.new_states(1)[0]
    scratch = np.zeros(2 * L, dtype=np.int16)
    want, no, lsum, lmax, lmaxmax = [], 1, 0.0, 0, 0
    import math
    for b in range(nb):
        k = lib.orc_block(C.byref(raw), C.byref(st), np.ascontiguousarray(iq[b * L:(b + 1) * L]), L, scratch)
        sr = lib.orc_rms(scratch.ctypes.data, k, 1, 0)
        no -= 1
        lsum += sr; lmax = max(lmax, sr); lmaxmax = max(lmaxmax, sr)
        if no == 0:
            no = n_every
            avg = lsum / n_every
            want.append("%.3f kHz, %.1f avg rms, %d max rms, %d max max rms, %d squelch rms, %d rms, %.1f dB rms level, %.2f dB avg rms level"
                        % (100000.0, avg, lmax, lmaxmax, 0, sr, 20 * math.log10(1e-10 + sr), 20 * math.log10(1e-10 + avg)))
            lmax, lsum = 0, 0.0
    assert lines == want
    # the PCM is what it is without -L
    got = np.fromfile(out, dtype=np.int16)
    ref, _ = oracle_lib.run_stream(cfg, iq)
    assert got.shape == ref.shape and np.abs(got.astype(np.int32) - ref.astype(np.int32)).max() <= 1


def test_ingest_bench_runs(tmp_path):
    """The native harness of bench.py's e2e leg (host/ingest_bench.cpp): T threads push, the main
    thread runs and fetches; every run must bring back the whole audio of every stream."""
    import json
    from rtlsdr_amd.capi import RtlfmCfg
    hipbuild.build_host()
    exe = os.path.join(os.path.dirname(hipbuild.CLI_OUT), "ingest_bench")
    cfg = RtlfmCfg.default(downsample=16, downsample_passes=4, rate_out=150000, block_len=65536, max_blocks=1)
    f = tmp_path / "c.cfg"
    f.write_bytes(bytes(cfg))
    p = subprocess.run([exe, str(f), "24", "5", "0.3"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["runs"] >= 3 and d["streams"] == 24 and d["threads"] == 5
    assert d["pcm_per_run"] == 24 * (65536 // 2 // 16)
