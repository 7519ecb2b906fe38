"""Python access to the CPU checker (oracle/liboracle.so) and, when present,
to the reference's own compiled hot path (oracle/_ref/libref_rtlfm.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (rtlsdr_amd) never imports
this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rtlsdr_amd.capi import RtlfmCfg, RtlfmStreamState, RtlpowerCfg

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
LOADER_SO = os.path.join(_HERE, "libref_loader.so")
REF_FM_SO = os.path.join(_HERE, "_ref", "libref_rtlfm.so")
REF_POWER_SO = os.path.join(_HERE, "_ref", "libref_rtlpower.so")

_P = C.POINTER
_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    """Compile the checker (and oracle/_ref when /root/reference exists).
    Everything make prints goes to stderr: bench.py's stdout is one JSON line."""
    import sys
    cmd = ["make", "-C", _HERE, "-s", "all"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=sys.stderr, stderr=sys.stderr)


_oracle = None


def oracle() -> C.CDLL:
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build()
        lib = C.CDLL(ORACLE_SO)
        lib.orc_state_init.argtypes = [_P(RtlfmStreamState)]
        lib.orc_block.argtypes = [_P(RtlfmCfg), _P(RtlfmStreamState), _u8p, C.c_uint32, _i16p]
        lib.orc_block.restype = C.c_int
        lib.orc_run_batch.argtypes = [
            _P(RtlfmCfg), _P(RtlfmStreamState), C.c_int, C.c_void_p, C.c_size_t, C.c_int,
            C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        lib.orc_run_batch.restype = C.c_int
        lib.orc_result_cap.argtypes = [_P(RtlfmCfg)]
        lib.orc_result_cap.restype = C.c_int
        lib.orc_fifth_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_generic_fir.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.orc_rotate16_neg90.argtypes = [C.c_void_p, C.c_int]
        lib.orc_rotate_90_u8.argtypes = [C.c_void_p, C.c_int]
        lib.orc_u8_to_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for n in ("orc_polar_discriminant", "orc_polar_disc_fast", "orc_polar_disc_lut"):
            getattr(lib, n).argtypes = [C.c_int] * 4
            getattr(lib, n).restype = C.c_int
        lib.orc_atan_lut.restype = _P(C.c_int32)
        lib.orc_low_pass.argtypes = [C.c_void_p, C.c_int, C.c_int, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]
        lib.orc_low_pass.restype = C.c_int
        lib.orc_deemph.argtypes = [C.c_void_p, C.c_int, C.c_int, _P(C.c_int32)]
        lib.orc_low_pass_real.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_int32), _P(C.c_int32)]
        lib.orc_low_pass_real.restype = C.c_int
        lib.orc_arbitrary_upsample.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.orc_arbitrary_downsample.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.orc_rms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.orc_rms.restype = C.c_int
        lib.orc_optimal_settings.argtypes = [
            _P(RtlfmCfg), C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
            _P(C.c_uint32), _P(C.c_uint32)]
        lib.orc_deemph_a.argtypes = [C.c_int, C.c_int]
        lib.orc_deemph_a.restype = C.c_int
        _oracle = lib
    return _oracle


def new_states(n: int):
    arr = (RtlfmStreamState * n)()
    for i in range(n):
        oracle().orc_state_init(C.byref(arr[i]))
    return arr


def result_cap(cfg: RtlfmCfg) -> int:
    return oracle().orc_result_cap(C.byref(cfg))


def run_stream(cfg: RtlfmCfg, iq: np.ndarray, state: RtlfmStreamState | None = None):
    """All blocks of ONE stream through the oracle.  Returns (int16 array, state)."""
    lib = oracle()
    iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
    L = int(cfg.block_len)
    assert iq.size % L == 0
    nb = iq.size // L
    if state is None:
        state = new_states(1)[0]
    cap = result_cap(cfg)
    out = np.zeros(cap * nb, dtype=np.int16)
    total = 0
    scratch = np.zeros(cap, dtype=np.int16)
    for b in range(nb):
        n = lib.orc_block(C.byref(cfg), C.byref(state), iq[b * L:(b + 1) * L], L, scratch)
        if n < 0:
            raise RuntimeError(f"orc_block -> {n}")
        out[total:total + n] = scratch[:n]
        total += n
    return out[:total].copy(), state


def run_batch(cfg: RtlfmCfg, iq: np.ndarray, states=None, nthreads: int = 1):
    """iq: uint8 [nstreams, nblocks*block_len].  Returns (out[nstreams, cap], out_len, states)."""
    lib = oracle()
    iq = np.ascontiguousarray(iq, dtype=np.uint8)
    ns = iq.shape[0]
    L = int(cfg.block_len)
    nb = iq.shape[1] // L
    if states is None:
        states = new_states(ns)
    cap = result_cap(cfg) * nb
    out = np.zeros((ns, cap), dtype=np.int16)
    out_len = np.zeros(ns, dtype=np.int32)
    r = lib.orc_run_batch(C.byref(cfg), states, ns, iq.ctypes.data, iq.strides[0], nb,
                          out.ctypes.data, cap, out_len.ctypes.data, nthreads)
    if r < 0:
        raise RuntimeError(f"orc_run_batch -> {r}")
    return out, out_len, states


# --------------------------------------------------------------------------- #
# the reference itself (only where oracle/_ref was built)
# --------------------------------------------------------------------------- #

def have_reference() -> bool:
    return os.path.exists(REF_FM_SO) and os.path.exists(LOADER_SO)


class Reference:
    """One freshly loaded copy of the reference's rtl_fm hot path.

    The reference keeps ONE stream in globals (and deemph_filter's avg in a
    function-static), so every instance dlopens a private copy of the file.
    """

    _count = 0

    def __init__(self, so_path: str = REF_FM_SO):
        import shutil
        import tempfile
        self._loader = C.CDLL(LOADER_SO)
        self._loader.ref_loader_open.restype = C.c_void_p
        self._loader.ref_loader_open.argtypes = [C.c_char_p]
        self._loader.ref_loader_close.argtypes = [C.c_void_p]
        # a private copy => a distinct dlopen identity => fresh globals/statics
        Reference._count += 1
        self._tmp = tempfile.NamedTemporaryFile(
            prefix=f"ref{os.getpid()}_{Reference._count}_", suffix=".so", delete=False)
        self._tmp.close()
        shutil.copyfile(so_path, self._tmp.name)
        self._handle = self._loader.ref_loader_open(self._tmp.name.encode())
        if not self._handle:
            raise OSError(f"cannot open {so_path}")
        self.lib = C.CDLL(self._tmp.name, handle=self._handle)
        L = self.lib
        L.ref_configure.argtypes = [_P(RtlfmCfg)]
        L.ref_configure.restype = C.c_int
        L.ref_block.argtypes = [_u8p, C.c_uint32, _i16p]
        L.ref_block.restype = C.c_int
        L.ref_run_stream.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.ref_run_stream.restype = C.c_int
        L.ref_state_get.argtypes = [_P(RtlfmStreamState)]
        L.ref_optimal_settings.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, _i32p]
        L.ref_atan_lut.restype = _P(C.c_int)
        L.ref_rotate_90_u8.argtypes = [C.c_void_p, C.c_uint32]
        L.ref_rotate_90_u8.restype = None
        L.ref_reset()

    def close(self):
        if getattr(self, "_handle", None):
            self._loader.ref_loader_close(self._handle)
            self._handle = None
            try:
                os.unlink(self._tmp.name)
            except OSError:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, cfg: RtlfmCfg):
        self.lib.ref_reset()
        if self.lib.ref_configure(C.byref(cfg)) != 0:
            raise ValueError("ref_configure")

    def run_stream(self, cfg: RtlfmCfg, iq: np.ndarray):
        """Fresh state, all blocks of one stream.  Returns (int16 array, state)."""
        self.configure(cfg)
        iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
        L = int(cfg.block_len)
        nb = iq.size // L
        cap = result_cap(cfg)
        out = np.zeros(cap * nb + 16, dtype=np.int16)
        n = self.lib.ref_run_stream(iq.ctypes.data, L, nb, out.ctypes.data)
        if n < 0:
            raise RuntimeError(f"ref_run_stream -> {n}")
        st = RtlfmStreamState()
        self.lib.ref_state_get(C.byref(st))
        return out[:n].copy(), st

    def optimal_settings(self, freq, rate_in, min_capture_rate, use_fifth, edge=0,
                         mode=0, offset_tuning=0):
        o = np.zeros(6, dtype=np.int32)
        self.lib.ref_optimal_settings(freq, rate_in, min_capture_rate, use_fifth, edge,
                                      mode, offset_tuning, o)
        return dict(downsample=int(o[0]), downsample_passes=int(o[1]), output_scale=int(o[2]),
                    capture_freq=int(np.uint32(o[3])), capture_rate=int(np.uint32(o[4])))


# --------------------------------------------------------------------------- #
# rtl_power
# --------------------------------------------------------------------------- #

def _power_lib():
    lib = oracle()
    if not hasattr(lib, "_power_ready"):
        lib.orcp_scan_batch.argtypes = [_P(RtlpowerCfg), C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int]
        lib.orcp_scan_batch.restype = C.c_int
        lib.orcp_window_coefs.argtypes = [C.c_int, C.c_int, C.c_void_p]
        lib.orcp_fix_fft.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.orcp_fix_fft.restype = C.c_int
        lib.orcp_sine_table.argtypes = [C.c_int]
        lib.orcp_sine_table.restype = C.c_void_p
        lib.orcp_fifth_order.argtypes = [C.c_void_p, C.c_int]
        lib.orcp_generic_fir.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.orcp_remove_dc.argtypes = [C.c_void_p, C.c_int]
        lib._power_ready = True
    return lib


def power_scan_batch(cfg: RtlpowerCfg, iq: np.ndarray, avg=None, samples=None, nthreads: int = 1):
    """iq: uint8 [nstreams, nreads*buf_len]; accumulates like scanner() does.
    Returns (avg int64 [nstreams, 2^bin_e], samples int32 [nstreams])."""
    lib = _power_lib()
    iq = np.ascontiguousarray(iq, dtype=np.uint8)
    ns = iq.shape[0]
    nreads = iq.shape[1] // int(cfg.buf_len)
    bins = 1 << cfg.bin_e
    if avg is None:
        avg = np.zeros((ns, bins), dtype=np.int64)
        samples = np.zeros(ns, dtype=np.int32)
    lib.orcp_scan_batch(C.byref(cfg), ns, iq.ctypes.data, iq.strides[0], nreads, avg.ctypes.data,
                        samples.ctypes.data, nthreads)
    return avg, samples


def power_window_coefs(window: int, length: int) -> np.ndarray:
    out = np.zeros(length, dtype=np.int32)
    _power_lib().orcp_window_coefs(window, length, out.ctypes.data)
    return out


def have_power_reference() -> bool:
    return os.path.exists(REF_POWER_SO) and os.path.exists(LOADER_SO)


class PowerReference:
    """A private copy of the reference's rtl_power DSP (oracle/_ref/libref_rtlpower.so)."""

    def __init__(self):
        import shutil
        import tempfile
        # the reference's scanner() calls into librtlsdr: the product's file-backed device layer
        # (26 rtlsdr_* symbols over a raw IQ file) is what those calls bind to
        from rtlsdr_amd import build as product_build
        C.CDLL(product_build.build_shim(), mode=C.RTLD_GLOBAL)
        self._loader = C.CDLL(LOADER_SO)
        self._loader.ref_loader_open.restype = C.c_void_p
        self._loader.ref_loader_open.argtypes = [C.c_char_p]
        self._loader.ref_loader_close.argtypes = [C.c_void_p]
        self._tmp = tempfile.NamedTemporaryFile(prefix=f"refp{os.getpid()}_", suffix=".so", delete=False)
        self._tmp.close()
        shutil.copyfile(REF_POWER_SO, self._tmp.name)
        self._handle = self._loader.ref_loader_open(self._tmp.name.encode())
        if not self._handle:
            raise OSError("cannot open " + REF_POWER_SO)
        self.lib = C.CDLL(self._tmp.name, handle=self._handle)
        self.lib.ref_power_setup.argtypes = [_P(RtlpowerCfg)]
        self.lib.ref_power_scan_file.argtypes = [C.c_char_p, C.c_int]
        self.lib.ref_power_get.argtypes = [C.c_void_p, _P(C.c_int32)]
        self.lib.ref_window_coefs.restype = _P(C.c_int)
        self.lib.ref_sinewave.restype = _P(C.c_int16)
        self.lib.fix_fft.argtypes = [C.c_void_p, C.c_int]
        self.lib.fifth_order.argtypes = [C.c_void_p, C.c_int]
        self.lib.remove_dc.argtypes = [C.c_void_p, C.c_int]
        self.lib.generic_fir.argtypes = [C.c_void_p, C.c_int, C.c_void_p]

    def close(self):
        if getattr(self, "_handle", None):
            self._loader.ref_loader_close(self._handle)
            self._handle = None
            try:
                os.unlink(self._tmp.name)
            except OSError:
                pass

    def scan_stream(self, cfg: RtlpowerCfg, iq: np.ndarray):
        """All reads of ONE stream; returns (avg int64 [2^bin_e], samples)."""
        self.lib.ref_power_setup(C.byref(cfg))
        import tempfile
        iq = np.ascontiguousarray(iq, dtype=np.uint8).ravel()
        L = int(cfg.buf_len)
        with tempfile.NamedTemporaryFile(prefix="refp_iq_", suffix=".bin") as f:
            iq[:iq.size // L * L].tofile(f)
            f.flush()
            r = self.lib.ref_power_scan_file(f.name.encode(), iq.size // L)
            assert r == 0, "the reference's scanner() could not open the file-backed device"
        avg = np.zeros(1 << cfg.bin_e, dtype=np.int64)
        n = C.c_int32()
        self.lib.ref_power_get(avg.ctypes.data, C.byref(n))
        return avg, n.value
