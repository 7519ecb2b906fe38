"""Host-side mirror of rtl_fm's per-buffer interface over the C ABI.

Names follow the reference (src/rtl_fm.c): ``rtlsdr_callback`` is what
``rtlsdr_read_async`` calls with a u8 IQ buffer (:1274), ``full_demod`` turns
the queued buffers into PCM (:1179), and what the output thread would
``fwrite`` (:1400) comes back from ``fetch``.  All arithmetic happens in the
HIP library; this class only moves pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import RtlfmCfg, RtlfmStreamState, check


class GpuDemod:
    """``nstreams`` independent rtl_fm demodulators sharing one configuration."""

    def __init__(self, cfg: RtlfmCfg, nstreams: int = 1, device: int = 0, lib_path: str | None = None,
                 options: dict | None = None):
        self.lib = capi.load(lib_path)  # lib_path: another build of the library (A/B measurements)
        self.cfg = cfg
        self.nstreams = nstreams
        self.device = device
        h = C.c_void_p()
        check(self.lib.rtlfm_gpu_create(C.byref(cfg), nstreams, device, C.byref(h)), "rtlfm_gpu_create")
        self._h = h
        for k, v in (options or {}).items():
            self.set_option(k, v)

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.rtlfm_gpu_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the callback boundary ---------------------------------------------
    def rtlsdr_callback(self, buf, stream: int = 0):
        """What rtlsdr_read_async's callback does with (buf, len, ctx)."""
        a = np.ascontiguousarray(buf, dtype=np.uint8)
        check(self.lib.rtlfm_gpu_push(self._h, stream, a.ctypes.data, a.size), "rtlfm_gpu_push")

    push = rtlsdr_callback

    def full_demod(self):
        check(self.lib.rtlfm_gpu_run(self._h), "rtlfm_gpu_run")

    run = full_demod

    def run_begin(self) -> int:
        """rtlfm_gpu_run_begin: take what the ring's filling half holds and flip the halves; returns buffers per stream taken."""
        n = C.c_int()
        check(self.lib.rtlfm_gpu_run_begin(self._h, C.byref(n)), "rtlfm_gpu_run_begin")
        return n.value

    def run_end(self):
        """rtlfm_gpu_run_end: the transfer and the kernels of the run that _begin took."""
        check(self.lib.rtlfm_gpu_run_end(self._h), "rtlfm_gpu_run_end")

    def fetch(self, stream: int = 0) -> np.ndarray:
        cap = capi.load().rtlfm_result_cap(C.byref(self.cfg)) * max(1, self.cfg.max_blocks) + 16
        out = np.empty(cap, dtype=np.int16)
        n = C.c_int()
        check(self.lib.rtlfm_gpu_fetch(self._h, stream, out.ctypes.data, cap, C.byref(n)), "rtlfm_gpu_fetch")
        return out[:n.value].copy()

    def fetch_all(self):
        """(out int16 [nstreams, cap], lens int32 [nstreams]) of the last full_demod(): one transfer."""
        cap = capi.load().rtlfm_result_cap(C.byref(self.cfg)) * max(1, self.cfg.max_blocks) + 16
        out = np.empty((self.nstreams, cap), dtype=np.int16)
        lens = np.zeros(self.nstreams, dtype=np.int32)
        check(self.lib.rtlfm_gpu_fetch_all(self._h, out.ctypes.data, cap, lens.ctypes.data), "rtlfm_gpu_fetch_all")
        return out, lens

    def fetch_all_prev(self):
        """The same for the run BEFORE the last one (rtlfm_gpu_fetch_all_prev): does not wait for the run started since."""
        cap = capi.load().rtlfm_result_cap(C.byref(self.cfg)) * max(1, self.cfg.max_blocks) + 16
        out = np.empty((self.nstreams, cap), dtype=np.int16)
        lens = np.zeros(self.nstreams, dtype=np.int32)
        check(self.lib.rtlfm_gpu_fetch_all_prev(self._h, out.ctypes.data, cap, lens.ctypes.data), "rtlfm_gpu_fetch_all_prev")
        return out, lens

    # -- device-resident form ------------------------------------------------
    def result_cap(self, nblocks: int) -> int:
        c = self.lib.rtlfm_result_cap(C.byref(self.cfg)) * nblocks + 16
        return (c + 63) & ~63  # rows start on 128-byte lines: the front end's tile stores then cover whole lines

    def run_device(self, d_iq_ptr: int, stream_stride: int, nblocks: int, d_out_ptr: int,
                   out_stride: int, d_out_len_ptr: int = 0):
        check(self.lib.rtlfm_gpu_run_device(self._h, d_iq_ptr, stream_stride, nblocks, d_out_ptr,
                                            out_stride, d_out_len_ptr or None), "rtlfm_gpu_run_device")

    def run_torch(self, iq, out=None, out_len=None):
        """iq: torch uint8 [nstreams, nblocks*block_len] on this device.
        Returns (out int16 [nstreams, cap], out_len int32 [nstreams]).

        Stream contract: the library launches on the handle's own stream.  This method orders it
        after torch's current stream (whatever produced ``iq`` and allocated / zeroed the outputs)
        and makes torch's current stream wait for the library's kernels before it returns, so the
        results can be used from torch without ``sync()`` and the caching allocator cannot hand
        ``out`` to someone else while the kernels still write it.  ``run_device`` does neither."""
        import torch
        assert iq.dtype == torch.uint8 and iq.is_cuda and iq.dim() == 2 and iq.shape[0] == self.nstreams
        assert iq.stride(1) == 1
        L = int(self.cfg.block_len)
        nb = iq.shape[1] // L
        cap = self.result_cap(nb)
        if out is None:
            out = torch.empty((self.nstreams, cap), dtype=torch.int16, device=iq.device)
        if out_len is None:
            out_len = torch.zeros(self.nstreams, dtype=torch.int32, device=iq.device)
        ts = torch.cuda.current_stream(iq.device).cuda_stream or None
        check(self.lib.rtlfm_gpu_wait_for(self._h, ts), "rtlfm_gpu_wait_for")
        self.run_device(iq.data_ptr(), iq.stride(0), nb, out.data_ptr(), out.stride(0), out_len.data_ptr())
        check(self.lib.rtlfm_gpu_release_to(self._h, ts), "rtlfm_gpu_release_to")
        return out, out_len

    def levels(self, stream: int = 0) -> np.ndarray:
        """rms() of the decimated IQ per buffer of the last run (needs squelch_level or report_levels)."""
        out = np.zeros(max(1, self.cfg.max_blocks), dtype=np.int32)
        n = C.c_int()
        check(self.lib.rtlfm_gpu_levels(self._h, stream, out.ctypes.data, out.size, C.byref(n)), "rtlfm_gpu_levels")
        return out[:n.value].copy()

    # -- state & plumbing ------------------------------------------------------
    def state_get(self, stream: int = 0) -> RtlfmStreamState:
        st = RtlfmStreamState()
        check(self.lib.rtlfm_gpu_state_get(self._h, stream, C.byref(st)), "rtlfm_gpu_state_get")
        return st

    def state_set(self, stream: int, st: RtlfmStreamState):
        check(self.lib.rtlfm_gpu_state_set(self._h, stream, C.byref(st)), "rtlfm_gpu_state_set")

    def reset(self):
        check(self.lib.rtlfm_gpu_reset(self._h), "rtlfm_gpu_reset")

    def sync(self):
        check(self.lib.rtlfm_gpu_sync(self._h), "rtlfm_gpu_sync")

    def set_stream(self, hip_stream_ptr: int):
        check(self.lib.rtlfm_gpu_set_stream(self._h, hip_stream_ptr or None), "rtlfm_gpu_set_stream")

    def set_option(self, name: str, value: int):
        """rtlfm_gpu_set_option: tunables and A/B switches by name (include/rtlfm_hip.h)."""
        check(self.lib.rtlfm_gpu_set_option(self._h, name.encode(), int(value)), f"rtlfm_gpu_set_option({name})")

    def get_option(self, name: str) -> int:
        v = C.c_long()
        check(self.lib.rtlfm_gpu_get_option(self._h, name.encode(), C.byref(v)), f"rtlfm_gpu_get_option({name})")
        return v.value

    def clock_stamps(self):
        """uint64 [waves, 4] of the last stamped launch: shader clock first / last, 100 MHz counter first / last."""
        n = C.c_int()
        if self.lib.rtlfm_gpu_clock_stamps(self._h, None, 0, C.byref(n)) < 0:
            return None
        out = np.zeros((n.value, 4), dtype=np.uint64)
        check(self.lib.rtlfm_gpu_clock_stamps(self._h, out.ctypes.data, n.value, C.byref(n)), "rtlfm_gpu_clock_stamps")
        return out

    def set_path(self, path: int):
        check(self.lib.rtlfm_gpu_set_path(self._h, path), "rtlfm_gpu_set_path")

    @property
    def last_path(self) -> int:
        return self.lib.rtlfm_gpu_last_path(self._h)

    def timing_enable(self, on: bool = True):
        check(self.lib.rtlfm_gpu_timing_enable(self._h, int(on)), "rtlfm_gpu_timing_enable")

    def clock_probe(self, on: bool = True):
        check(self.lib.rtlfm_gpu_clock_probe(self._h, int(on)), "rtlfm_gpu_clock_probe")

    def clock_read(self):
        """(mean shader MHz, span ms) of the last fused fifth_order launch, or None."""
        mhz, span = C.c_double(), C.c_double()
        r = self.lib.rtlfm_gpu_clock_read(self._h, C.byref(mhz), C.byref(span))
        if r < 0:
            return None
        return mhz.value, span.value

    def timing_read(self):
        ms = C.c_double()
        n = C.c_int()
        check(self.lib.rtlfm_gpu_timing_read(self._h, C.byref(ms), C.byref(n)), "rtlfm_gpu_timing_read")
        return ms.value, n.value


def optimal_settings(cfg: RtlfmCfg, freq: int, rate_in: int, min_capture_rate: int = 1000000,
                     use_fifth_order: bool = False, edge: int = 0):
    """optimal_settings() (src/rtl_fm.c:1407-1445); returns (capture_freq, capture_rate)."""
    cf, cr = C.c_uint32(), C.c_uint32()
    check(capi.load().rtlfm_optimal_settings(C.byref(cfg), freq, rate_in, min_capture_rate,
                                             int(use_fifth_order), edge, C.byref(cf), C.byref(cr)),
          "rtlfm_optimal_settings")
    return cf.value, cr.value


def deemph_a(rate_out: int, time_constant_us: int = 75) -> int:
    return capi.load().rtlfm_deemph_a(rate_out, time_constant_us)
