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


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """The gfx950 assembly of both translation units, compiled once (side by side) for every lint of this file."""
    td = tmp_path_factory.mktemp("isa")
    procs = []
    for u in ("rtlfm_hip", "rtlpower_hip", "rtlfm_place"):
        out = str(td / (u + ".s"))
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
               os.path.join(ROOT, "rtlsdr_amd", "csrc", u + ".hip"), "-o", out]
        procs.append((out, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    outs = []
    for out, p in procs:
        assert p.wait(timeout=900) == 0, out
        outs.append(out)
    return outs


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_nothing_touches_a_dot_result_inside_its_hazard_window(device_asm):
    """gfx90a and later do not interlock v_dot2 / v_dot4 results against the next VALU instructions; the compiler keeps the
    distance (3 wait states before a read, 2 before a write by another opcode) for its own dots, not for inline assembly.
    Round 5 shipped 37 kernels whose bare v_dot2_i32_i16 wrapper was read one and two states later (LAB.md I.20)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dot_hazard.py"), *device_asm],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.endswith("0 function(s) with a dot hazard"), last
    assert int(last.split(" functions, ")[1].split(" dot")[0]) > 5000, last  # the lint did see the kernels' dots


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_fused_kernels_keep_their_prefetch_in_flight(device_asm):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch.py"), device_asm[0]],
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
