"""rtlsdr_amd — MI355X (gfx950) implementation of rtl_fm's demod hot path.

The product is the C-ABI shared library built from ``rtlsdr_amd/csrc`` (see
``include/rtlfm_hip.h``).  The Python in this package is the host-side mirror
of the reference's per-buffer interface (``rtlsdr_callback`` -> ``full_demod``
-> output) over that ABI, plus build and synthetic-signal helpers.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
__version__ = "0.1.0"
