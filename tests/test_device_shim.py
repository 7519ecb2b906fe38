"""The file-backed rtlsdr_* device layer (SURVEY.md §8b "device shim"): exports the
26 symbols rtl_fm / rtl_power / convenience link against, and behaves like the
reference's reader where a tool can observe it (src/librtlsdr.c:2826-2952)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rtlsdr_amd import build as hipbuild

SYMS = """rtlsdr_cancel_async rtlsdr_close rtlsdr_get_center_freq rtlsdr_get_device_count rtlsdr_get_device_name
rtlsdr_get_device_usb_strings rtlsdr_get_tuner_gains rtlsdr_get_ver_id rtlsdr_get_version rtlsdr_open
rtlsdr_read_async rtlsdr_read_sync rtlsdr_reset_buffer rtlsdr_set_agc_mode rtlsdr_set_and_get_tuner_bandwidth
rtlsdr_set_bias_tee rtlsdr_set_center_freq rtlsdr_set_direct_sampling rtlsdr_set_ds_mode
rtlsdr_set_freq_correction_ppb rtlsdr_set_offset_tuning rtlsdr_set_opt_string rtlsdr_set_sample_rate
rtlsdr_set_tuner_bandwidth rtlsdr_set_tuner_gain rtlsdr_set_tuner_gain_mode""".split()

CB = C.CFUNCTYPE(None, C.POINTER(C.c_ubyte), C.c_uint32, C.c_void_p)


@pytest.fixture(scope="module")
def shim():
    so, _ = hipbuild.build_host()
    lib = C.CDLL(so)
    lib.rtlsdr_open.argtypes = [C.POINTER(C.c_void_p), C.c_uint32]
    lib.rtlsdr_close.argtypes = [C.c_void_p]
    lib.rtlsdr_read_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.rtlsdr_read_async.argtypes = [C.c_void_p, CB, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.rtlsdr_cancel_async.argtypes = [C.c_void_p]
    lib.rtlsdr_set_center_freq.argtypes = [C.c_void_p, C.c_uint32]
    lib.rtlsdr_get_center_freq.argtypes = [C.c_void_p]
    lib.rtlsdr_get_center_freq.restype = C.c_uint32
    lib.rtlsdr_set_sample_rate.argtypes = [C.c_void_p, C.c_uint32]
    lib.rtlsdr_get_tuner_gains.argtypes = [C.c_void_p, C.c_void_p]
    lib.rtlsdr_get_device_name.restype = C.c_char_p
    lib.rtlsdr_get_device_name.argtypes = [C.c_uint32]
    return lib, so


def test_exports_exactly_the_26_symbols(shim):
    _, so = shim
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    have = sorted(l.split()[-1] for l in out.splitlines() if " T rtlsdr_" in l)
    assert have == sorted(SYMS) and len(have) == 26


def test_async_reads_into_consumer_memory(shim, tmp_path, monkeypatch):
    """The zero-copy extension (rtlamd_file_set_buffer_source): rtlsdr_read_async reads every buffer
    straight into memory the consumer names and hands the callback that pointer - what rtl_fm_hip -Z does
    with the GPU layer's pinned ring slots (the reference's counterpart: use_zerocopy,
    src/librtlsdr.c:2744-2810)."""
    lib, _ = shim
    data = np.random.default_rng(2).integers(0, 256, size=16384 * 4 + 700, dtype=np.uint8)
    f = tmp_path / "iq.bin"
    data.tofile(f)
    monkeypatch.setenv("RTLSDR_FILE", str(f))
    h = C.c_void_p()
    assert lib.rtlsdr_open(C.byref(h), 0) == 0
    mine = np.zeros((6, 16384), dtype=np.uint8)
    handed, seen = [], []
    SRC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32))

    def source(ctx, buf, cap):
        k = len(handed)
        if k == 2:          # "no slot this time": the library falls back to its own buffer
            handed.append(None)
            return -1
        handed.append(mine[k].ctypes.data)
        buf[0] = mine[k].ctypes.data
        cap[0] = 16384
        return 0

    def cb(buf, n, ctx):
        seen.append((C.cast(buf, C.c_void_p).value, n, bytes(C.string_at(buf, n))))
    src_fn, cb_fn = SRC(source), CB(cb)
    lib.rtlamd_file_set_buffer_source.argtypes = [C.c_void_p, SRC, C.c_void_p]
    assert lib.rtlamd_file_set_buffer_source(h, src_fn, None) == 0
    assert lib.rtlsdr_read_async(h, cb_fn, None, 0, 16384) == 0
    assert [n for _, n, _ in seen] == [16384] * 4 + [700]
    assert b"".join(b for _, _, b in seen) == data.tobytes()
    for k, (ptr, n, _) in enumerate(seen):
        if handed[k] is None:
            assert ptr not in [x for x in handed if x]        # its own buffer
        else:
            assert ptr == handed[k]                            # read in place, no copy
            assert bytes(mine[k][:n]) == data[k * 16384:k * 16384 + n].tobytes()
    lib.rtlsdr_close(h)


def test_a_slot_that_is_too_small_is_given_back(shim, tmp_path, monkeypatch):
    """A buffer source may offer less room than a transfer needs: the device layer must hand that slot back (the callback
    sees the slot's pointer with 0 bytes, the owner commits nothing) before it reads into a buffer of its own - round 3
    left the slot open, and a ring whose slot stays open never runs again."""
    lib, _ = shim
    data = np.random.default_rng(3).integers(0, 256, size=16384 * 3, dtype=np.uint8)
    f = tmp_path / "iq.bin"
    data.tofile(f)
    monkeypatch.setenv("RTLSDR_FILE", str(f))
    h = C.c_void_p()
    assert lib.rtlsdr_open(C.byref(h), 0) == 0
    small = np.zeros(4096, dtype=np.uint8)
    seen = []
    SRC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32))

    def source(ctx, buf, cap):
        buf[0] = small.ctypes.data
        cap[0] = 4096          # a transfer is 16384 bytes
        return 0

    def cb(buf, n, ctx):
        seen.append((C.cast(buf, C.c_void_p).value, n, bytes(C.string_at(buf, n))))
    src_fn, cb_fn = SRC(source), CB(cb)
    lib.rtlamd_file_set_buffer_source.argtypes = [C.c_void_p, SRC, C.c_void_p]
    assert lib.rtlamd_file_set_buffer_source(h, src_fn, None) == 0
    assert lib.rtlsdr_read_async(h, cb_fn, None, 0, 16384) == 0
    given_back = [x for x in seen if x[0] == small.ctypes.data]
    own = [x for x in seen if x[0] != small.ctypes.data]
    assert len(given_back) >= 3 and all(n == 0 for _, n, _ in given_back)
    assert b"".join(b for _, _, b in own) == data.tobytes()
    lib.rtlsdr_close(h)


def test_no_file_means_no_device(shim, monkeypatch):
    lib, _ = shim
    monkeypatch.delenv("RTLSDR_FILE", raising=False)
    assert lib.rtlsdr_get_device_count() == 0
    h = C.c_void_p()
    assert lib.rtlsdr_open(C.byref(h), 0) < 0
    assert lib.rtlsdr_read_async(None, CB(lambda *a: None), None, 0, 0) == -1  # NULL device


def test_sync_and_async_reads(shim, tmp_path, monkeypatch):
    lib, _ = shim
    data = np.random.default_rng(1).integers(0, 256, size=16384 * 5 + 1000, dtype=np.uint8)
    f = tmp_path / "iq.bin"
    data.tofile(f)
    monkeypatch.setenv("RTLSDR_FILE", str(f))
    assert lib.rtlsdr_get_device_count() == 1 and b"file" in lib.rtlsdr_get_device_name(0)
    h = C.c_void_p()
    assert lib.rtlsdr_open(C.byref(h), 0) == 0
    assert lib.rtlsdr_set_center_freq(h, 99400000) == 0 and lib.rtlsdr_get_center_freq(h) == 99400000
    assert lib.rtlsdr_set_sample_rate(h, 2400000) == 0
    assert lib.rtlsdr_set_sample_rate(h, 500000) < 0  # the RTL2832's invalid window (src/librtlsdr.c:1633-1637)
    gains = (C.c_int * 64)()
    assert lib.rtlsdr_get_tuner_gains(h, gains) > 10
    buf = (C.c_ubyte * 4096)(); n = C.c_int()
    assert lib.rtlsdr_read_sync(h, buf, 4096, C.byref(n)) == 0 and n.value == 4096
    assert bytes(buf) == data[:4096].tobytes()
    got = []

    def cb(p, ln, ctx):
        got.append(bytes(C.cast(p, C.POINTER(C.c_ubyte * ln)).contents))
    assert lib.rtlsdr_read_async(h, CB(cb), None, 0, 16384) == 0  # returns at end of file
    assert b"".join(got) == data[4096:].tobytes()
    assert [len(g) for g in got[:-1]] == [16384] * (len(got) - 1) and len(got[-1]) < 16384
    assert lib.rtlsdr_cancel_async(h) == -2  # nothing running
    lib.rtlsdr_close(h)


def test_cancel_from_callback_and_wav_header(shim, tmp_path, monkeypatch):
    lib, _ = shim
    payload = np.arange(65536, dtype=np.uint32).astype(np.uint8)
    hdr = b"RIFF" + (36 + payload.size).to_bytes(4, "little") + b"WAVE" + b"fmt " + (16).to_bytes(4, "little") + \
        (1).to_bytes(2, "little") + (2).to_bytes(2, "little") + (2400000).to_bytes(4, "little") + \
        (4800000).to_bytes(4, "little") + (2).to_bytes(2, "little") + (8).to_bytes(2, "little") + \
        b"data" + payload.size.to_bytes(4, "little")
    f = tmp_path / "iq.wav"
    f.write_bytes(hdr + payload.tobytes())
    monkeypatch.setenv("RTLSDR_FILE", str(f))
    monkeypatch.setenv("RTLSDR_FILE_LOOP", "1")
    h = C.c_void_p()
    assert lib.rtlsdr_open(C.byref(h), 0) == 0
    seen = []

    def cb(p, ln, ctx):
        seen.append(bytes(C.cast(p, C.POINTER(C.c_ubyte * ln)).contents))
        if len(seen) == 7:
            lib.rtlsdr_cancel_async(h)
    # buf_len not a multiple of 512 falls back to 16*32*512 (src/librtlsdr.c:2853-2855)
    assert lib.rtlsdr_read_async(h, CB(cb), None, 3, 1000) == 0
    assert len(seen) == 7 and all(len(s) == 262144 for s in seen)
    assert seen[0][:16] == payload[:16].tobytes()  # header skipped; loops over the 64 KiB payload
    lib.rtlsdr_close(h)


def test_wave_header_matches_reference_writer(shim, tmp_path, oracle_lib):
    """§8f-2: the 120-byte RIFF/fmt/auxi/data header, field for field against the
    reference's own wavewrite.c (compiled into oracle/_ref/libref_rtlfm.so); the two
    SYSTEMTIME stamps are wall-clock and only checked for plausibility."""
    lib, _ = shim
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    libc.fwrite.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    lib.rtlamd_wave_write_header_file.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_void_p]
    lib.rtlamd_wave_finalize_file.argtypes = [C.c_void_p]
    payload = bytes(range(200)) * 7
    mine = tmp_path / "mine.wav"
    f = libc.fopen(str(mine).encode(), b"wb")
    lib.rtlamd_wave_write_header_file(32000, 99400000, 16, 1, f)
    libc.fwrite(payload, 1, len(payload), f)
    lib.rtlamd_wave_add_data(len(payload))
    lib.rtlamd_wave_finalize_file(f)
    libc.fclose(f)
    a = mine.read_bytes()
    assert len(a) == 120 + len(payload) and a[:4] == b"RIFF" and a[36:40] == b"auxi" and a[112:116] == b"data"
    assert int.from_bytes(a[4:8], "little") == 112 + len(payload)
    assert int.from_bytes(a[116:120], "little") == len(payload)
    assert int.from_bytes(a[76:80], "little") == 99400000 and int.from_bytes(a[32:34], "little") == 1
    assert 2024 <= int.from_bytes(a[44:46], "little") <= 2100
    if not oracle_lib.have_reference():
        return
    ref = oracle_lib.Reference()
    try:
        ref.lib.waveWriteHeader.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_void_p]
        ref.lib.waveFinalizeHeader.argtypes = [C.c_void_p]
        theirs = tmp_path / "ref.wav"
        f = libc.fopen(str(theirs).encode(), b"wb")
        ref.lib.waveWriteHeader(32000, 99400000, 16, 1, f)
        libc.fwrite(payload, 1, len(payload), f)
        C.c_uint32.in_dll(ref.lib, "waveDataSize").value = len(payload)
        ref.lib.waveFinalizeHeader(f)
        libc.fclose(f)
        b = theirs.read_bytes()
    finally:
        ref.close()
    assert len(a) == len(b)
    assert a[:44] == b[:44] and a[76:] == b[76:]  # everything but the two time stamps


def test_rtl_tcp_source(shim, monkeypatch):
    """§8f-4: RTLSDR_FILE=tcp://host:port speaks rtl_tcp's wire format (protocol_rtl_tcp.txt):
    12-byte "RTL0" dongle_info, then raw u8 IQ; setters become 5-byte big-endian commands."""
    import socket
    import threading
    lib, _ = shim
    payload = np.random.default_rng(4).integers(0, 256, size=16384 * 3 + 700, dtype=np.uint8).tobytes()
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)
    port = srv.getsockname()[1]
    commands = []

    def serve():
        c, _ = srv.accept()
        c.sendall(b"RTL0" + (5).to_bytes(4, "big") + (29).to_bytes(4, "big"))  # R820T, 29 gains
        c.settimeout(2.0)
        buf = b""
        try:
            while len(buf) < 15:  # three commands
                buf += c.recv(64)
        except OSError:
            pass
        commands.extend(buf[i:i + 5] for i in range(0, len(buf) - len(buf) % 5, 5))
        c.sendall(payload)
        c.close()
    t = threading.Thread(target=serve, daemon=True)
    t.start()
    monkeypatch.setenv("RTLSDR_FILE", f"tcp://127.0.0.1:{port}")
    h = C.c_void_p()
    assert lib.rtlsdr_open(C.byref(h), 0) == 0
    assert lib.rtlsdr_set_center_freq(h, 99400000) == 0
    assert lib.rtlsdr_set_sample_rate(h, 2400000) == 0
    lib.rtlsdr_set_agc_mode.argtypes = [C.c_void_p, C.c_int]
    assert lib.rtlsdr_set_agc_mode(h, 1) == 0
    got = []

    def cb(p, ln, ctx):
        got.append(bytes(C.cast(p, C.POINTER(C.c_ubyte * ln)).contents))
    assert lib.rtlsdr_read_async(h, CB(cb), None, 0, 16384) == 0  # ends when the server closes
    t.join(5)
    lib.rtlsdr_close(h)
    srv.close()
    assert b"".join(got) == payload
    assert commands[0] == b"\x01" + (99400000).to_bytes(4, "big")
    assert commands[1] == b"\x02" + (2400000).to_bytes(4, "big")
    assert commands[2] == b"\x08" + (1).to_bytes(4, "big")
