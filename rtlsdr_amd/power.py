"""Host-side mirror of rtl_power's scanner() over the C ABI (include/rtlpower_hip.h).

One ``GpuPower`` handle holds ``nstreams`` tuning states (reference
``struct tuning_state``, src/rtl_power.c:86-108): ``scanner`` feeds reads,
``avg``/``samples`` are what ``csv_dbm`` (:722-765) reads."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import RtlpowerCfg, check


class GpuPower:
    def __init__(self, cfg: RtlpowerCfg, nstreams: int = 1, device: int = 0):
        self.lib = capi.load()
        self.cfg = cfg
        self.nstreams = nstreams
        h = C.c_void_p()
        check(self.lib.rtlpower_gpu_create(C.byref(cfg), nstreams, device, C.byref(h)), "rtlpower_gpu_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.rtlpower_gpu_destroy(self._h)
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

    def scanner(self, buf, stream: int = 0):
        """One rtlsdr_read_sync() buffer of one tuning state."""
        a = np.ascontiguousarray(buf, dtype=np.uint8)
        check(self.lib.rtlpower_gpu_scan(self._h, stream, a.ctypes.data, a.size), "rtlpower_gpu_scan")

    def scan_device(self, d_iq_ptr: int, stream_stride: int, nreads: int):
        check(self.lib.rtlpower_gpu_scan_device(self._h, d_iq_ptr, stream_stride, nreads),
              "rtlpower_gpu_scan_device")

    def scan_torch(self, iq):
        """iq: torch uint8 [nstreams, nreads*buf_len] on the device.  Ordered after torch's current
        stream (the producer of ``iq``), and torch's current stream waits for the scan, so ``iq``
        may be freed or overwritten from torch right away."""
        import torch
        nreads = iq.shape[1] // int(self.cfg.buf_len)
        ts = torch.cuda.current_stream(iq.device).cuda_stream or None
        check(self.lib.rtlpower_gpu_wait_for(self._h, ts), "rtlpower_gpu_wait_for")
        self.scan_device(iq.data_ptr(), iq.stride(0), nreads)
        check(self.lib.rtlpower_gpu_release_to(self._h, ts), "rtlpower_gpu_release_to")

    def fetch(self, stream: int = 0):
        n = 1 << self.cfg.bin_e
        avg = np.zeros(n, dtype=np.int64)
        samples = C.c_int32()
        check(self.lib.rtlpower_gpu_fetch(self._h, stream, avg.ctypes.data, C.byref(samples)), "rtlpower_gpu_fetch")
        return avg, samples.value

    def clear(self):
        check(self.lib.rtlpower_gpu_clear(self._h), "rtlpower_gpu_clear")

    def set_option(self, name: str, value: int):
        """Tunables by name (include/rtlpower_hip.h): "groups", "staged_fast", "scan_frames", "dec_fast"."""
        check(self.lib.rtlpower_gpu_set_option(self._h, name.encode(), int(value)), f"rtlpower_gpu_set_option({name})")

    def get_option(self, name: str) -> int:
        v = C.c_long()
        check(self.lib.rtlpower_gpu_get_option(self._h, name.encode(), C.byref(v)), f"rtlpower_gpu_get_option({name})")
        return v.value

    @property
    def last_kernel(self) -> int:
        """Which transform the last scan took (RTLPOWER_KERNEL_*: 1 general, 2 big, 3 frames, 4 decimated, 5 / 6 staged)."""
        return self.get_option("last_kernel")

    def sync(self):
        check(self.lib.rtlpower_gpu_sync(self._h), "rtlpower_gpu_sync")

    def timing_enable(self, on=True):
        check(self.lib.rtlpower_gpu_timing_enable(self._h, int(on)), "timing_enable")

    def timing_read(self):
        ms, n = C.c_double(), C.c_int()
        check(self.lib.rtlpower_gpu_timing_read(self._h, C.byref(ms), C.byref(n)), "timing_read")
        return ms.value, n.value


    def clock_probe(self, on=True):
        check(self.lib.rtlpower_gpu_clock_probe(self._h, int(on)), "rtlpower_gpu_clock_probe")

    def clock_read(self):
        """(mean shader MHz of the last launch's workgroups, first-start-to-last-end span in ms) or None."""
        mhz, span = C.c_double(), C.c_double()
        r = self.lib.rtlpower_gpu_clock_read(self._h, C.byref(mhz), C.byref(span))
        return (mhz.value, span.value) if r == 0 else None


def window_coefs(window: int, length: int) -> np.ndarray:
    out = np.zeros(length, dtype=np.int32)
    check(capi.load().rtlpower_window_coefs(window, length, out.ctypes.data), "rtlpower_window_coefs")
    return out
