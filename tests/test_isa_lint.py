"""The generated gfx950 ISA of the fused kernels keeps the next tile's loads in flight.

hipcc cross-compiles without a GPU, so this runs in the CPU suite.  It guards a
performance property that no parity test can see: a register-allocator copy of a
freshly loaded register puts an HBM round trip into every tile (13 % on the headline
workload when it happened).
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_fused_kernels_keep_their_prefetch_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    # every instantiation: 6 pass counts x FIR on/off x std / run-time discriminator x two engines,
    # and the MFMA engine once more with the raw DC block (-E rdc) on its accumulators
    ok_main = sum(1 for ln in r.stdout.splitlines() if ln.endswith("pt=0: ok"))
    assert ok_main == 72, r.stdout[-3000:]
    assert "STALL" not in r.stdout
    # ... and the partial-tile kernels (-W n, MFMA engine): 24 of them, nearly all clean (a by-road; reported only)
    ok_pt = sum(1 for ln in r.stdout.splitlines() if ln.endswith("pt=1: ok"))
    assert ok_pt >= 20, r.stdout[-3000:]
