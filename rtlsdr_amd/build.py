"""Build the HIP product for gfx950 with hipcc (in-tree, so the .so travels
with the repository snapshot to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "librtlfm_hip.so")
SOURCES = ["rtlfm_hip.hip", "rtlpower_hip.hip", "rtlfm_place.hip"]
HEADERS = ["dsp_device.h", "staged_kernels.h", "fused_kernel.h",
           os.path.join("..", "..", "include", "rtlfm_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError("hipcc not found")


OBJDIR = os.path.join(CSRC, "build")
# which headers a translation unit sees (a header change recompiles only the units that include it)
UNIT_HEADERS = {
    "rtlfm_hip.hip": ["debug_poison.h", "stream_pool.h", "dsp_device.h", "staged_kernels.h", "fused_kernel.h", "boxcar_kernel.h",
                      os.path.join("..", "..", "include", "rtlfm_hip.h")],
    "rtlfm_place.hip": ["debug_poison.h", "bw_probe_kernel.h", os.path.join("..", "..", "include", "rtlfm_hip.h")],
    "rtlpower_hip.hip": ["debug_poison.h", "stream_pool.h", "dsp_device.h", "power_kernels.h", os.path.join("..", "..", "include", "rtlpower_hip.h"),
                         os.path.join("..", "..", "include", "rtlfm_hip.h")],
}
# -ffile-prefix-map: __FILE__ (HIP_TRY's messages) and every other path the compiler embeds are written relative to the
# repository, so that two checkouts in different directories build the same bytes (a rebuild under /tmp
# differed from the shipped library in its path strings)
ROOT = os.path.dirname(HERE)
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value",
         f"-ffile-prefix-map={ROOT}=."]


def _obj(src: str) -> str:
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")


def _flags_key(extra) -> str:
    return " ".join(FLAGS + list(extra or []))


def _unit_stale(src: str, extra=None) -> bool:
    """An object is reused only if it is newer than everything it was built from AND was built with the same flags
    (an A/B build with -D... must not leave its objects to a later plain build)."""
    o = _obj(src)
    if not os.path.exists(o):
        return True
    try:
        with open(o + ".flags") as f:
            if f.read() != _flags_key(extra):
                return True
    except OSError:
        return True
    t = os.path.getmtime(o)
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in UNIT_HEADERS[src]]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra: list[str] | None = None) -> str:
    """One object per translation unit (compiled side by side), then one link: librtlfm_hip.so.  Objects and the
    library are written under a temporary name and renamed, and the whole build holds a file lock: several test
    processes that find the library stale at once (pytest -n) neither link nor load half-written files."""
    if not force and not extra and not needs_build():
        return OUT
    import fcntl
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(OBJDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not extra and not needs_build():
            return OUT  # another process built it while this one waited
        procs = []
        tag = f".tmp{os.getpid()}"
        for src in SOURCES:
            if not (force or _unit_stale(src, extra)):
                continue
            # -cuid: the compilation unit's id is otherwise a hash that takes the (temporary, per-process) output name in -
            # two builds of the same sources then differ in their symbol names and the library in its bytes
            cmd = ([_hipcc()] + FLAGS + [f"-cuid={os.path.splitext(src)[0]}", "-c", "-o", _obj(src) + tag, os.path.join(CSRC, src)]
                   + (extra or []))
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, cmd, subprocess.Popen(cmd)))
        for src, cmd, p in procs:
            if p.wait() != 0:
                raise subprocess.CalledProcessError(p.returncode, cmd)
            os.replace(_obj(src) + tag, _obj(src))
            with open(_obj(src) + ".flags", "w") as f:
                f.write(_flags_key(extra))
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", "-o", OUT + tag] + [_obj(s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(OUT + tag, OUT)
    return OUT


HOST = os.path.join(CSRC, "host")
SHIM_OUT = os.path.join(HOST, "librtlsdr_file.so")
CLI_OUT = os.path.join(HOST, "rtl_fm_hip")
POWER_CLI_OUT = os.path.join(HOST, "rtl_power_hip")
INGEST_OUT = os.path.join(HOST, "ingest_bench")


def build_shim(force: bool = False, verbose: bool = False) -> str:
    """Only the file-backed librtlsdr device layer (plain gcc; no GPU library needed)."""
    inc = os.path.join(CSRC, "..", "..", "include")
    shim_src = os.path.join(HOST, "rtlsdr_file.c")
    deps = [shim_src, os.path.join(inc, "rtlsdr_file.h")]
    if force or not os.path.exists(SHIM_OUT) or any(os.path.getmtime(x) > os.path.getmtime(SHIM_OUT) for x in deps):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", SHIM_OUT, shim_src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return SHIM_OUT


def build_host(force: bool = False, verbose: bool = False) -> tuple[str, str]:
    """The file-backed librtlsdr device layer (26 rtlsdr_* symbols) and the rtl_fm-shaped
    CLI that sits on it and on librtlfm_hip.so.  Plain gcc/g++."""
    def stale(out, srcs):
        return force or not os.path.exists(out) or any(os.path.getmtime(x) > os.path.getmtime(out) for x in srcs)
    inc = os.path.join(CSRC, "..", "..", "include")
    shim_src = os.path.join(HOST, "rtlsdr_file.c")
    if stale(SHIM_OUT, [shim_src, os.path.join(inc, "rtlsdr_file.h")]):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", SHIM_OUT, shim_src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    cli_src = os.path.join(HOST, "rtl_fm_hip.cpp")
    if stale(CLI_OUT, [cli_src, SHIM_OUT, OUT, os.path.join(inc, "rtlfm_hip.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", CLI_OUT, cli_src,
               "-L" + HOST, "-L" + CSRC, "-lrtlsdr_file", "-lrtlfm_hip", "-lpthread",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib",
               "-Wl,--allow-shlib-undefined"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    pw_src = os.path.join(HOST, "rtl_power_hip.cpp")
    if stale(POWER_CLI_OUT, [pw_src, SHIM_OUT, OUT, os.path.join(inc, "rtlpower_hip.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", POWER_CLI_OUT, pw_src,
               "-L" + HOST, "-L" + CSRC, "-lrtlsdr_file", "-lrtlfm_hip", "-lm",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib",
               "-Wl,--allow-shlib-undefined"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    ib_src = os.path.join(HOST, "ingest_bench.cpp")
    if stale(INGEST_OUT, [ib_src, OUT, os.path.join(inc, "rtlfm_hip.h")]):
        # the PCIe-inclusive rate of the callback boundary from native threads (bench.py's e2e leg)
        cmd = ["g++", "-O2", "-std=c++20", "-Wall", "-o", INGEST_OUT, ib_src, "-L" + CSRC, "-lrtlfm_hip", "-lpthread",
               "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return SHIM_OUT, CLI_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
