"""ctypes binding of the C ABI declared in ``include/rtlfm_hip.h``.

The shared library (``rtlsdr_amd/csrc/librtlfm_hip.so``) is the product; this
module only marshals plain pointers and sizes into it.  There is no CPU
fallback: if the library is missing, or it finds no GPU, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librtlfm_hip.so")

RTLFM_MAX_PASSES = 10
RTLFM_MAX_BLOCK_LEN = 262144

MODE_FM, MODE_AM, MODE_USB, MODE_LSB, MODE_RAW = range(5)
ATAN_STD, ATAN_FAST, ATAN_LUT = range(3)
RESAMPLE_LOW_PASS_REAL, RESAMPLE_ARBITRARY = range(2)
PATH_AUTO, PATH_STAGED, PATH_FUSED = range(3)


class RtlfmCfg(C.Structure):
    """``rtlfm_cfg`` — the demod_state fields that stay constant while
    samples flow (reference: src/rtl_fm.c:172-208)."""

    _fields_ = [
        ("mode", C.c_int32),
        ("downsample", C.c_int32),
        ("downsample_passes", C.c_int32),
        ("comp_fir_size", C.c_int32),
        ("custom_atan", C.c_int32),
        ("post_downsample", C.c_int32),
        ("deemph", C.c_int32),
        ("deemph_a", C.c_int32),
        ("rate_out", C.c_int32),
        ("rate_out2", C.c_int32),
        ("resampler", C.c_int32),
        ("dc_block_audio", C.c_int32),
        ("adc_block_const", C.c_int32),
        ("dc_block_raw", C.c_int32),
        ("rdc_block_const", C.c_int32),
        ("offset_tuning", C.c_int32),
        ("output_scale", C.c_int32),
        ("squelch_level", C.c_int32),
        ("block_len", C.c_uint32),
        ("max_blocks", C.c_int32),
        ("report_levels", C.c_int32),
    ]

    @classmethod
    def from_saved(cls, raw: bytes) -> "RtlfmCfg":
        """A configuration stored by an earlier round's fixture: later fields are zero."""
        return cls.from_buffer_copy(raw.ljust(C.sizeof(cls), b"\0"))

    @classmethod
    def default(cls, **kw) -> "RtlfmCfg":
        """demod_init() defaults (src/rtl_fm.c:1608-1640) plus overrides."""
        c = cls()
        c.mode = MODE_FM
        c.downsample = 1
        c.downsample_passes = 0
        c.comp_fir_size = 0
        c.custom_atan = ATAN_STD
        c.post_downsample = 1
        c.deemph = 0
        c.deemph_a = 0
        c.rate_out = 24000
        c.rate_out2 = -1
        c.resampler = RESAMPLE_LOW_PASS_REAL
        c.dc_block_audio = 0
        c.adc_block_const = 9
        c.dc_block_raw = 0
        c.rdc_block_const = 9
        c.offset_tuning = 0
        c.output_scale = 1
        c.squelch_level = 0
        c.block_len = 16384
        c.max_blocks = 1
        for k, v in kw.items():
            if not hasattr(c, k):
                raise AttributeError(k)
            setattr(c, k, v)
        return c

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


class RtlfmStreamState(C.Structure):
    """``rtlfm_stream_state`` — what one stream carries between blocks."""

    _fields_ = [
        ("lp_i_hist", (C.c_int16 * 6) * RTLFM_MAX_PASSES),
        ("lp_q_hist", (C.c_int16 * 6) * RTLFM_MAX_PASSES),
        ("droop_i_hist", C.c_int16 * 9),
        ("droop_q_hist", C.c_int16 * 9),
        ("pad_", C.c_int16 * 2),
        ("now_r", C.c_int32),
        ("now_j", C.c_int32),
        ("prev_index", C.c_int32),
        ("pre_r", C.c_int32),
        ("pre_j", C.c_int32),
        ("now_lpr", C.c_int32),
        ("prev_lpr_index", C.c_int32),
        ("deemph_avg", C.c_int32),
        ("dc_avg", C.c_int32),
        ("dc_avgI", C.c_int32),
        ("dc_avgQ", C.c_int32),
        ("squelch_hits", C.c_int32),
    ]

    def as_dict(self) -> dict:
        d = {}
        for n, _ in self._fields_:
            if n == "pad_":
                continue
            v = getattr(self, n)
            if n in ("lp_i_hist", "lp_q_hist"):
                v = [list(r) for r in v]
            elif hasattr(v, "__len__"):
                v = list(v)
            d[n] = v
        return d


WIN_RECTANGLE, WIN_HAMMING, WIN_BLACKMAN, WIN_BLACKMAN_HARRIS, WIN_HANN_POISSON, WIN_YOUSSEF, WIN_KAISER, \
    WIN_BARTLETT = range(8)


class RtlpowerCfg(C.Structure):
    """``rtlpower_cfg`` — what scanner() reads (reference src/rtl_power.c:86-120)."""

    _fields_ = [
        ("bin_e", C.c_int32),
        ("window", C.c_int32),
        ("downsample", C.c_int32),
        ("downsample_passes", C.c_int32),
        ("boxcar", C.c_int32),
        ("comp_fir_size", C.c_int32),
        ("peak_hold", C.c_int32),
        ("buf_len", C.c_uint32),
    ]

    @classmethod
    def default(cls, **kw) -> "RtlpowerCfg":
        c = cls(bin_e=10, window=WIN_RECTANGLE, downsample=1, downsample_passes=0, boxcar=1,
                comp_fir_size=0, peak_hold=0, buf_len=16384)
        for k, v in kw.items():
            if not hasattr(c, k):
                raise AttributeError(k)
            setattr(c, k, v)
        return c

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


# Every symbol include/rtlfm_hip.h declares: (name, restype, argtypes)
_P = C.POINTER
_SIGNATURES = [
    ("rtlfm_cfg_default", None, [_P(RtlfmCfg)]),
    ("rtlfm_optimal_settings", C.c_int,
     [_P(RtlfmCfg), C.c_uint32, C.c_int32, C.c_int32, C.c_int, C.c_int,
      _P(C.c_uint32), _P(C.c_uint32)]),
    ("rtlfm_deemph_a", C.c_int32, [C.c_int32, C.c_int32]),
    ("rtlfm_result_len", C.c_int, [_P(RtlfmCfg)]),
    ("rtlfm_result_cap", C.c_int, [_P(RtlfmCfg)]),
    ("rtlfm_cfg_validate", C.c_int, [_P(RtlfmCfg)]),
    ("rtlfm_gpu_create", C.c_int, [_P(RtlfmCfg), C.c_int, C.c_int, _P(C.c_void_p)]),
    ("rtlfm_gpu_destroy", C.c_int, [C.c_void_p]),
    ("rtlfm_gpu_push", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]),
    ("rtlfm_gpu_acquire", C.c_int, [C.c_void_p, C.c_int, _P(C.c_void_p), _P(C.c_uint32)]),
    ("rtlfm_gpu_commit", C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    ("rtlfm_gpu_run", C.c_int, [C.c_void_p]),
    ("rtlfm_gpu_run_begin", C.c_int, [C.c_void_p, _P(C.c_int)]),
    ("rtlfm_gpu_run_end", C.c_int, [C.c_void_p]),
    ("rtlfm_gpu_run_device", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rtlfm_gpu_fetch", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, _P(C.c_int)]),
    ("rtlfm_gpu_fetch_all", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rtlfm_gpu_fetch_all_prev", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rtlfm_gpu_levels", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, _P(C.c_int)]),
    ("rtlfm_gpu_state_get", C.c_int, [C.c_void_p, C.c_int, _P(RtlfmStreamState)]),
    ("rtlfm_gpu_state_set", C.c_int, [C.c_void_p, C.c_int, _P(RtlfmStreamState)]),
    ("rtlfm_gpu_reset", C.c_int, [C.c_void_p]),
    ("rtlfm_gpu_sync", C.c_int, [C.c_void_p]),
    ("rtlfm_gpu_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rtlfm_gpu_wait_for", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rtlfm_gpu_release_to", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rtlfm_gpu_set_path", C.c_int, [C.c_void_p, C.c_int]),
    ("rtlfm_gpu_last_path", C.c_int, [C.c_void_p]),
    ("rtlfm_gpu_timing_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("rtlfm_gpu_timing_read", C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_int)]),
    ("rtlfm_gpu_clock_probe", C.c_int, [C.c_void_p, C.c_int]),
    ("rtlfm_gpu_clock_read", C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_double)]),
    ("rtlfm_gpu_clock_stamps", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _P(C.c_int)]),
    ("rtlfm_gpu_bw_probe", C.c_int, [C.c_int, C.c_size_t, C.c_int, C.c_int, _P(C.c_double), _P(C.c_double), _P(C.c_double),
                                     _P(C.c_double)]),
    ("rtlfm_gpu_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_long]),
    ("rtlfm_gpu_get_option", C.c_int, [C.c_void_p, C.c_char_p, _P(C.c_long)]),
    ("rtlfm_gpu_selftest_atan2", C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("rtlfm_gpu_selftest_fast_atan2", C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    ("rtlfm_gpu_selftest_const_div", C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    ("rtlfm_gpu_rotate_90_u8", C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rtlfm_gpu_malloc", C.c_int, [C.c_int, C.c_size_t, _P(C.c_void_p)]),
    ("rtlfm_gpu_malloc_apart", C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, _P(C.c_void_p), _P(C.c_int)]),
    ("rtlfm_gpu_malloc_apart_ex", C.c_int, [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, _P(C.c_void_p), _P(C.c_int),
                                            _P(C.c_double), _P(C.c_size_t)]),
    ("rtlfm_gpu_placement_probe", C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _P(C.c_double), _P(C.c_double)]),
    ("rtlfm_gpu_place_pair", C.c_int, [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _P(C.c_void_p), _P(C.c_void_p), _P(C.c_int), _P(C.c_int),
                                       _P(C.c_double), _P(C.c_size_t)]),
    ("rtlfm_gpu_copy", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("rtlfm_gpu_free", C.c_int, [C.c_void_p]),
    ("rtlfm_gpu_device_numa_node", C.c_int, [C.c_int]),
    ("rtlfm_plan_segments", C.c_int, [C.c_int] * 8 + [_P(C.c_int), _P(C.c_int), C.c_int]),
    ("rtlfm_gpu_strerror", C.c_char_p, [C.c_int]),
    ("rtlfm_gpu_version", C.c_int, []),
]

class RtlpowerPlan(C.Structure):
    """``rtlpower_plan`` — frequency_range()'s hop plan (reference src/rtl_power.c:438-540)."""

    _fields_ = [("lower", C.c_int32), ("upper", C.c_int32), ("max_size", C.c_int32), ("tune_count", C.c_int32),
                ("bw_seen", C.c_int32), ("rate", C.c_int32), ("bin_e", C.c_int32), ("downsample", C.c_int32),
                ("downsample_passes", C.c_int32), ("buf_len", C.c_int32), ("crop", C.c_double),
                ("bin_size", C.c_double)]

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


# ... and include/rtlpower_hip.h
_POWER_SIGNATURES = [
    ("rtlpower_frequency_range", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int, _P(RtlpowerPlan)]),
    ("rtlpower_tune_freq", C.c_int32, [_P(RtlpowerPlan), C.c_int]),
    ("rtlpower_plan_cfg", None, [_P(RtlpowerPlan), C.c_int, C.c_int, C.c_int, C.c_int, _P(RtlpowerCfg)]),
    ("rtlpower_csv_dbm", C.c_int, [_P(RtlpowerPlan), C.c_int, C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t]),
    ("rtlpower_window_coefs", C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    ("rtlpower_cfg_validate", C.c_int, [_P(RtlpowerCfg)]),
    ("rtlpower_gpu_create", C.c_int, [_P(RtlpowerCfg), C.c_int, C.c_int, _P(C.c_void_p)]),
    ("rtlpower_gpu_destroy", C.c_int, [C.c_void_p]),
    ("rtlpower_gpu_scan_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    ("rtlpower_gpu_scan", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]),
    ("rtlpower_gpu_fetch", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _P(C.c_int32)]),
    ("rtlpower_gpu_clear", C.c_int, [C.c_void_p]),
    ("rtlpower_gpu_sync", C.c_int, [C.c_void_p]),
    ("rtlpower_gpu_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rtlpower_gpu_wait_for", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rtlpower_gpu_release_to", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rtlpower_gpu_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_long]),
    ("rtlpower_gpu_get_option", C.c_int, [C.c_void_p, C.c_char_p, _P(C.c_long)]),
    ("rtlpower_gpu_timing_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("rtlpower_gpu_timing_read", C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_int)]),
    ("rtlpower_gpu_clock_probe", C.c_int, [C.c_void_p, C.c_int]),
    ("rtlpower_gpu_clock_read", C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_double)]),
]
_SIGNATURES = _SIGNATURES + _POWER_SIGNATURES

DECLARED_SYMBOLS = [s[0] for s in _SIGNATURES]
DECLARED_FM_SYMBOLS = [s[0] for s in _SIGNATURES if s[0].startswith("rtlfm_")]
DECLARED_POWER_SYMBOLS = [s[0] for s in _POWER_SIGNATURES]

_lib = None


class RtlfmError(RuntimeError):
    def __init__(self, code: int, what: str):
        self.code = code
        super().__init__(f"{what}: {code} ({strerror(code)})")


def load(path: str | None = None) -> C.CDLL:
    """Load librtlfm_hip.so (once) and attach the prototypes."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RTLFM_HIP_LIB") or LIB_PATH  # env override: A/B builds
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 and
    # libhsa-runtime64 under the same sonames as /opt/rocm's.  If this library
    # is opened first it binds /opt/rocm's copy, torch then brings a second HSA
    # runtime and the device disappears for whoever comes second.  Importing
    # torch first (when it is installed) makes both share torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    for name, res, args in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def strerror(code: int) -> str:
    try:
        return load().rtlfm_gpu_strerror(code).decode()
    except Exception:
        return os.strerror(-code) if code < 0 else "ok"


def check(code: int, what: str) -> int:
    if code < 0:
        raise RtlfmError(code, what)
    return code
