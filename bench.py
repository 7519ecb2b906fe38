#!/usr/bin/env python3
"""bench.py — IQ Msamples/s demodulated on MI355X, with roofline and CPU baseline.

One "step" is one pass of the hot path (u8 IQ -> rotate -> decimate -> discriminator [-> audio
tail] -> int16 PCM, reference src/rtl_fm.c:1326-1338 + 1179-1272; for `--workload c4` rtl_power's
scanner(), src/rtl_power.c:642-720) over one batch of synthetic input that is already resident
in HBM.

    python bench.py [--workload c2] --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json `configs`; the default line is north_star's own shape of configs[1]'s chain, and every
other BASELINE configuration is timed by the same invocation as a short `also` leg):

    ns4096  north_star's target shape: 4096 batched 2.4 MS/s streams x 4 buffers x 262144 B, -F 0 (4x fifth_order,
            /16), -A std                                                                          [default]
    c2      configs[1] as BASELINE words it: 256 streams x 64 buffers x 262144 B through the same /16 FM path
    c1      config 0's chain batched: low_pass boxcar /10 + -A fast, 256 x 64 x 262144 B @2.4 MS/s
    wbfm    rtl_fm -M wbfm: boxcar /6, -A fast, deemph, low_pass_real 170k -> 32k, 1024 x 16 x 262144 B
    c3      4096 NBFM streams @1.024 MS/s: 6x fifth_order + FIR9 (/64), deemph, arbitrary_resample -> 22050
    c4      rtl_power: 1024 streams x 64 reads x 32768 B, hamming window + 16k-bin fix_fft + integrate

Streams are independent: rank r owns its own streams (weak scaling), there is no data-path
collective; RCCL is used only for the barrier and the max-over-ranks of the elapsed time.
`python bench.py --gpus N` outside torchrun starts the N ranks itself.

Besides the contract fields the line carries
  roofline.sustained   the same step repeated for >= --sustain seconds (the short K-step figure is
                       taken right after the warm-up; the GPU sits at its 1.4 kW cap on this path
                       and the clock settles over seconds)
  roofline.traffic     HBM bytes per launch of the dominant kernel from two rocprofv3 --pmc passes
                       (FETCH_SIZE, WRITE_SIZE; separate runs, no tracing) made by this very
                       invocation on this box before the timed run (N = 1; --pmc 0 skips them)
  roofline.shader_mhz  mean shader clock of the waves of a launch, stamped in the kernel
  roofline.ceiling     THIS box's streaming ceilings, measured by this invocation with the front end's access
                       pattern and none of its arithmetic (rtlfm_gpu_bw_probe): read only, and read + write at the
                       workload's own byte ratio; roofline.frac_of_ceiling = achieved / that read+write ceiling.
                       `frac` stays against the nominal 8 TB/s; the two together tell box from code.
  also                 (default workload) the other BASELINE shapes in the same invocation, each with its own
                       HIP-event launch_ms, wall ms_per_step, frac / step_frac and frac_of_ceiling: ns4096x1 (one
                       buffer per stream per launch: what a live capture hands over per callback round) and c2
                       on the resident bytes, then c3, c1, wbfm and c4 (valu_issue) on inputs of their own
  e2e                  host buffers through rtlfm_gpu_push / _run / _fetch_all (PCIe-inclusive; never `value`)
  cpu_baseline         the reference's own code (oracle/_ref) or the oracle port on the host cores
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

# several ranks on one node share device memory handles over dmabuf only (this image exports it already; a launcher
# that builds its own environment may not) - before anything loads the HIP runtime
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md

# name -> argument defaults (explicit flags win)
WORKLOADS = {
    "c2": dict(streams=256, blocks=64, block_len=262144, passes=4, boxcar=0, fir9=0, atan="std", fs=2.4e6, tail=""),
    "ns4096": dict(streams=4096, blocks=4, block_len=262144, passes=4, boxcar=0, fir9=0, atan="std", fs=2.4e6, tail=""),
    "c1": dict(streams=256, blocks=64, block_len=262144, passes=0, boxcar=10, fir9=0, atan="fast", fs=2.4e6, tail=""),
    "wbfm": dict(streams=1024, blocks=16, block_len=262144, passes=0, boxcar=6, fir9=0, atan="fast", fs=1.02e6, tail="wbfm"),
    "c3": dict(streams=4096, blocks=4, block_len=262144, passes=6, boxcar=0, fir9=1, atan="std", fs=1.024e6, tail="c3"),
    "c4": dict(streams=1024, blocks=64, block_len=32768, passes=0, boxcar=0, fir9=0, atan="std", fs=2.048e6, tail="power"),
    # configs[1]'s chain on the reference's OWN buffer size (dongle_init(), src/rtl_fm.c:1605: 16384 bytes): every second
    # tile starts a buffer
    "c2_16k": dict(streams=256, blocks=1024, block_len=16384, passes=4, boxcar=0, fir9=0, atan="std", fs=2.4e6, tail=""),
    # the everyday command line, rtl_fm -M fm -s 12k -l 50 (boxcar /84 at 1.008 MS/s + the power squelch)
    "scanner": dict(streams=256, blocks=64, block_len=262144, passes=0, boxcar=84, fir9=0, atan="std", fs=1.008e6, tail="", squelch=50),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a launch is ~0.9 ms; after idle the GPU needs ~50 of them to reach its steady clock
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="ns4096")
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU")
    ap.add_argument("--blocks", type=int, default=None, help="callback buffers (c4: reads) per stream per step")
    ap.add_argument("--block-len", type=int, default=None)
    ap.add_argument("--passes", type=int, default=None)
    ap.add_argument("--boxcar", type=int, default=None,
                    help="D > 0: the reference's default low_pass boxcar /D instead of fifth_order passes")
    ap.add_argument("--fir9", type=int, default=None)
    ap.add_argument("--rdc", type=int, default=0, help="1: -E rdc (dc_block_raw_filter) in front of the chain")
    ap.add_argument("--squelch", type=int, default=None, help="-l N: the power squelch behind the decimator (the front end's emit mode + squelch kernels)")
    ap.add_argument("--atan", choices=["std", "fast", "lut"], default=None)
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 staged, 2 fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--check", type=int, default=1, help="have the cpu_baseline leg compare a slice of the GPU output with the CPU path")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of the sustained leg after the K timed steps (0: skip)")
    ap.add_argument("--pmc", type=int, default=-1, help="rocprofv3 PMC passes for roofline.traffic: 1 on, 0 off, -1 on for N = 1 when rocprofv3 exists")
    ap.add_argument("--e2e", type=int, default=1, help="1: also time the PCIe-inclusive push / run / fetch path (N = 1)")
    ap.add_argument("--ceiling", type=int, default=1, help="1: measure this box's streaming ceilings (roofline.ceiling)")
    ap.add_argument("--colocate", type=int, default=0, help="1: leave the output wherever the allocator puts it (normally the input's HBM quarter)")
    ap.add_argument("--also", type=int, default=1, help="1: the default workload also times the other BASELINE shapes (short legs)")
    ap.add_argument("--also-steps", type=int, default=200, help="timed launches per `also` leg (after 100 untimed)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--scatter", action="store_true",
                    help="N > 1: also time shard.scatter_streams (root GPU -> its owner GPUs) of one step's IQ over RCCL")
    a = ap.parse_args()
    if a.boxcar is not None and a.boxcar > 0 and a.passes is None:
        a.passes = 0
    if a.passes is not None and a.passes > 0 and a.boxcar is None:
        a.boxcar = 0
    for k, v in WORKLOADS[a.workload].items():
        if getattr(a, k, None) is None:
            setattr(a, k, v)
    if a.squelch is None:
        a.squelch = 0
    if a.boxcar:
        a.passes, a.fir9 = 0, 0
    return a


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` outside torchrun: start one rank per GPU ourselves.

    Runs before anything in this process has touched the GPU: the ranks are children of a
    `python -m torch.distributed.run` subprocess (never a re-exec of a process that holds the
    device).  torch.cuda.device_count() only counts, it does not initialise the runtime here.
    The JSON line is printed by rank 0 of the child job; this process only relays its exit code.
    """
    import socket

    import torch
    have = torch.cuda.device_count()
    n = a.gpus
    shared = os.environ.get("RTLFM_BENCH_BACKEND", "nccl") != "nccl"  # gloo: ranks may share a device (control-flow tests)
    if have < 1:
        print("bench.py needs a GPU (no CPU fallback)", file=sys.stderr)
        return 2
    if n > have and not shared:
        print(f"bench.py: --gpus {n} asked, {have} visible: running {have} rank(s); n_gpus reports what joined",
              file=sys.stderr)
        n = have
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = list(sys.argv[1:])
    env = dict(os.environ, RTLFM_BENCH_REQUESTED_GPUS=str(a.gpus), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------- PMC passes ----

def dominant_kernel(a) -> str:
    if a.tail == "power":
        return "k_power_scan"
    return "k_boxcar_scan" if a.boxcar else "k_fused"


def pmc_counters(a, counters):
    """Per-launch averages of hardware counters for the dominant kernel of workload `a`, measured now, on this
    box: one child run of this script under `rocprofv3 --pmc` per counter (no tracing domains, as
    MI355X_MICROARCH.md prescribes), started before this process touches the GPU.  None if anything fails."""
    rp = shutil.which("rocprofv3")
    if not rp:
        return None
    import csv
    pat = dominant_kernel(a)
    base = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--pmc", "0", "--gpus", "1", "--steps", "3",
            "--warmup", "2", "--sustain", "0", "--no-cpu-baseline", "--check", "0", "--workload", a.workload,
            "--streams", str(a.streams), "--blocks", str(a.blocks), "--block-len", str(a.block_len),
            "--path", str(a.path), "--atan", a.atan, "--colocate", "1", "--ceiling", "0", "--also", "0", "--e2e", "0"]
    base += ["--boxcar", str(a.boxcar)] if a.boxcar else ["--passes", str(a.passes), "--fir9", str(a.fir9)]
    base += ["--rdc", str(a.rdc), "--squelch", str(getattr(a, "squelch", 0))]
    out = {}
    tmp = tempfile.mkdtemp(prefix="rtlfm_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for ctr in counters:
            d = os.path.join(tmp, ctr)
            cmd = [rp, "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + base
            t0 = time.perf_counter()
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                print(f"bench.py: rocprofv3 --pmc {ctr} did not finish in 240 s; left null", file=sys.stderr)
                return None
            print(f"bench.py: rocprofv3 --pmc {ctr} pass ({a.workload}) took {time.perf_counter() - t0:.0f} s", file=sys.stderr)
            if r.returncode != 0:
                print(f"bench.py: rocprofv3 --pmc {ctr} failed ({r.returncode}): {r.stderr[-300:]}", file=sys.stderr)
                return None
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if pat in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                            per.setdefault(row["Dispatch_Id"], 0.0)
                            per[row["Dispatch_Id"]] += float(row["Counter_Value"])
            if not per:
                return None
            big = [v for v in per.values() if v >= 0.5 * max(per.values())]  # the parity-gate sized launches are left out
            out[ctr] = sum(big) / len(big)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def pmc_traffic(a):
    """HBM bytes per launch of the dominant kernel.  FETCH_SIZE / WRITE_SIZE count KiB at the L2's memory side;
    on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming loads at 64 bytes, so it
    is doubled (MI355X_MICROARCH.md)."""
    out = pmc_counters(a, ("FETCH_SIZE", "WRITE_SIZE"))
    if not out:
        return None
    fetch_b, write_b = 2.0 * out["FETCH_SIZE"] * 1024.0, out["WRITE_SIZE"] * 1024.0
    return {"bytes": int(fetch_b + write_b), "fetch_bytes": int(fetch_b), "write_bytes": int(write_b),
            "kernel": dominant_kernel(a), "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this invocation on this box; "
                                  "FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 B), KiB -> bytes"}


def workload_args(a, name):
    """a copy of the parsed arguments with workload `name`'s shape"""
    import copy
    b = copy.copy(a)
    b.workload = name
    b.squelch = 0
    for k, v in WORKLOADS[name].items():
        setattr(b, k, v)
    b.rdc, b.path = 0, 0
    if b.boxcar:
        b.passes, b.fir9 = 0, 0
    return b


# --------------------------------------------------------------- CPU baseline ----

def cpu_baseline(cfg, iq_host_sample, seconds, gate=None):
    """CPU baseline on the GPU box's host cores, bounded to roughly `seconds`.

    kind "reference": the reference's own rtlsdr_callback() + full_demod(), compiled
    in place from /root/reference into oracle/_ref/libref_rtlfm.so (prebuilt file
    travels with the repo snapshot); one private copy of the library per thread
    because the reference keeps a single stream in globals.
    kind "port": the oracle's C restatement (pinned bit-exact to the reference),
    pthreads over streams — used when oracle/_ref is absent.
    """
    import ctypes as C

    import numpy as np
    from oracle import pyoracle as po
    if gate is not None:
        # parity of the measured product path with the CPU path on a slice of the same input
        gcfg, giq, gout, glen = gate
        want, wl, _ = po.run_batch(gcfg, giq, nthreads=4)
        assert (glen == wl).all(), "parity gate: output counts differ"
        for s in range(len(wl)):
            d = np.abs(gout[s, :wl[s]].astype(np.int32) - want[s, :wl[s]].astype(np.int32))
            assert d.max() <= 1 and (d != 0).mean() <= 1e-4, f"parity gate failed: max {d.max()}, {(d != 0).sum()} differ"
    cores = os.cpu_count() or 1
    ns, nbytes = iq_host_sample.shape
    L = int(cfg.block_len)
    nb = nbytes // L
    threads = min(cores, ns)
    if po.have_reference():
        ld = C.CDLL(po.LOADER_SO)
        ld.ref_bench_mt.restype = C.c_double
        ld.ref_bench_mt.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32,
                                    C.c_int, C.c_int, C.c_int]

        def run(nt, reps):
            return ld.ref_bench_mt(po.REF_FM_SO.encode(), C.byref(cfg), iq_host_sample.ctypes.data,
                                   iq_host_sample.strides[0], L, nb, nt, reps)
        t1 = run(1, 2)  # one thread, for the per-core figure
        tc = run(threads, 1)
        if t1 > 0 and tc > 0:
            reps = max(1, int(seconds / tc))
            dt = run(threads, reps)
            if dt > 0:
                samples = reps * threads * nb * (L // 2)
                return {
                    "value": round(samples / dt / 1e6, 2), "unit": "Msamples/s", "cores": threads,
                    "kind": "reference",
                    "sample": f"reference rtl_fm.c (gcc -O3) rtlsdr_callback+full_demod: {threads} threads x 1 stream "
                              f"x {nb} buffers x {L} B x {reps} reps ({samples / 1e6:.0f} Msamples in {dt:.1f} s); "
                              f"one thread alone: {2 * nb * (L // 2) / t1 / 1e6:.1f} Msamples/s",
                }
    # fallback: the port
    t0 = time.perf_counter()
    po.run_batch(cfg, iq_host_sample[:, :L].copy(), nthreads=threads)
    t_one = max(time.perf_counter() - t0, 1e-4)
    reps = max(1, int(seconds / (t_one * nb)))
    t0 = time.perf_counter()
    states = None
    for _ in range(reps):
        _, _, states = po.run_batch(cfg, iq_host_sample, states=states, nthreads=threads)
    dt = time.perf_counter() - t0
    samples = reps * ns * nb * (L // 2)
    return {
        "value": round(samples / dt / 1e6, 2), "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"oracle port: {ns} streams x {nb} buffers x {L} B x {reps} reps "
                  f"({samples / 1e6:.0f} Msamples, {dt:.1f} s, {threads} pthreads)",
    }


def cpu_baseline_power(cfg, sample, seconds, gate=None):
    """rtl_power's scanner() on the host cores, bounded.

    kind "reference": the reference's own scanner() (src/rtl_power.c:642-720) compiled in place into
    oracle/_ref/libref_rtlpower.so, one private copy of the library per thread (scanner() keeps its
    tuning state, FFT buffer and tables in globals), every thread on its own file-backed device that
    serves one looped capture (the product's librtlsdr_file.so is what scanner()'s rtlsdr_read_sync binds to).
    kind "port": the oracle's restatement, one pthread per stream - when oracle/_ref is absent."""
    import ctypes as C
    import tempfile

    import numpy as np
    from oracle import pyoracle as po
    if gate is not None:
        giq, gavg, gsamples = gate
        want, wn = po.power_scan_batch(cfg, giq, nthreads=4)
        assert np.array_equal(wn, gsamples) and np.array_equal(want, gavg), "parity gate failed (rtl_power)"
    L = int(cfg.buf_len)
    cores = os.cpu_count() or 1
    if po.have_power_reference():
        from rtlsdr_amd import build as product_build
        C.CDLL(product_build.SHIM_OUT, mode=C.RTLD_GLOBAL)  # the device layer scanner() reads through
        ld = C.CDLL(po.LOADER_SO)
        ld.ref_power_bench_mt.restype = C.c_double
        ld.ref_power_bench_mt.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        with tempfile.NamedTemporaryFile(prefix="rtlpower_cap_", suffix=".bin") as tf:
            sample[0].tofile(tf); tf.flush()
            os.environ["RTLSDR_FILE"] = tf.name
            os.environ["RTLSDR_FILE_LOOP"] = "1"
            try:
                t1 = ld.ref_power_bench_mt(po.REF_POWER_SO.encode(), C.byref(cfg), 1, 8)
                tc = ld.ref_power_bench_mt(po.REF_POWER_SO.encode(), C.byref(cfg), cores, 8)
                dt, nscans = -1.0, 0
                if t1 > 0 and tc > 0:
                    nscans = max(8, int(8 * seconds / tc))
                    dt = ld.ref_power_bench_mt(po.REF_POWER_SO.encode(), C.byref(cfg), cores, nscans)
            finally:
                os.environ.pop("RTLSDR_FILE", None)
                os.environ.pop("RTLSDR_FILE_LOOP", None)
        if dt > 0:
            samples = cores * nscans * (L // 2)
            return {"value": round(samples / dt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "reference",
                    "sample": f"reference rtl_power.c (gcc -O3) scanner(): {cores} threads x 1 tuning state x {nscans} reads x {L} B "
                              f"({samples / 1e6:.0f} Msamples in {dt:.1f} s); one thread alone: {8 * (L // 2) / t1 / 1e6:.1f} Msamples/s"}
    cs = sample.shape[0]
    t1 = time.perf_counter()
    po.power_scan_batch(cfg, sample, nthreads=cs)
    one = max(time.perf_counter() - t1, 1e-3)
    reps = max(1, int(seconds / one))
    t1 = time.perf_counter()
    for _ in range(reps):
        po.power_scan_batch(cfg, sample, nthreads=cs)
    dt = time.perf_counter() - t1
    nreads = sample.shape[1] // L
    samples = reps * cs * nreads * (L // 2)
    return {"value": round(samples / dt / 1e6, 2), "unit": "Msamples/s", "cores": cs, "kind": "port",
            "sample": f"oracle port of scanner() (src/rtl_power.c:642-720): {cs} pthreads x 1 stream x {nreads} reads x {L} B "
                      f"x {reps} reps ({samples / 1e6:.0f} Msamples in {dt:.1f} s)"}


def e2e_leg(a, job, local_rank, seconds=3.0, quick=False):
    """PCIe-inclusive rate through the callback boundary: host buffers -> rtlfm_gpu_push
    (memcpy into the pinned ring, 16 pushing threads as 16 dongle threads would) -> rtlfm_gpu_run (async
    H2D + kernels) -> rtlfm_gpu_fetch_all (one D2H), pipelined: the callbacks of run k + 1 fill the other
    half of the ring while run k is in flight.  Next to it the box's own pinned H2D rate for the same
    bytes (the ceiling this path can reach) — never `value`."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
    import torch
    from rtlsdr_amd.capi import RtlfmCfg
    from rtlsdr_amd.demod import GpuDemod
    S = min(a.streams, 1024)
    L = a.block_len
    cfg = RtlfmCfg.from_buffer_copy(bytes(job.cfg))
    cfg.max_blocks = 1
    nbytes = S * L

    def h2d_ceiling():  # the same bytes, pinned host -> device, nothing else
        pin = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        dst = torch.empty(nbytes, dtype=torch.uint8, device=job.iq.device)
        dst.copy_(pin, non_blocking=True); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            dst.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        return 5 * nbytes / (time.perf_counter() - t1) / 1e9

    # native threads (rtlsdr_amd/csrc/host/ingest_bench.cpp, a child process: nothing here execs);
    # the Python threads below top out on the interpreter lock at ~8 and are only the fall-back
    native = os.path.join(ROOT, "rtlsdr_amd", "csrc", "host", "ingest_bench")
    if os.path.exists(native) and not os.environ.get("RTLFM_E2E_PYTHON"):
        import subprocess
        import tempfile
        nthreads = int(os.environ.get("RTLFM_E2E_THREADS", "8"))  # 8: 46 GB/s, 16-128: 37-43 on the box it was tried on

        def run_native(c, streams, mode, secs):
            with tempfile.NamedTemporaryFile(suffix=".cfg") as tf:
                tf.write(bytes(c)); tf.flush()
                try:
                    p = subprocess.run([native, tf.name, str(streams), str(nthreads), str(secs), "--devices", str(local_rank),
                                        "--mode", mode], capture_output=True, text=True, timeout=180)
                    return json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else None
                except Exception:  # noqa: BLE001 - fall back to the Python threads
                    return None
        res = run_native(cfg, S, "push", seconds)
        if res:
            h2d = round(h2d_ceiling(), 1)
            out = {"value": res["Msamples/s"], "unit": "Msamples/s", "GB/s_in": res["GB/s_in"],
                   "pinned_h2d_GB/s": h2d, "frac_of_pinned_h2d": round(res["GB/s_in"] / h2d, 3), "harness": "native",
                   "what": f"{S} streams x 1 buffer x {L} B per run, {res['runs']} pipelined runs in {res['seconds']:.2f} s: {nthreads} "
                           f"native threads rtlfm_gpu_push (pageable -> pinned ring) | rtlfm_gpu_run (async H2D + kernels) | "
                           f"rtlfm_gpu_fetch_all_prev (two runs in flight: run k + 1 is started before the audio of run k is "
                           f"collected, so the H2D copies follow each other on the link); bounded by PCIe, not by the kernels"}
            if quick:  # the per-rank leg of an N > 1 run: the push loop and the H2D rate only
                return out
            za = run_native(cfg, S, "acquire", seconds)
            if za:
                out["zero_copy"] = {"GB/s_in": za["GB/s_in"], "value": za["Msamples/s"], "frac_of_pinned_h2d": round(za["GB/s_in"] / h2d, 3),
                                    "what": "the same loop with rtlfm_gpu_acquire / _commit: the producer writes the pinned ring slot itself "
                                            "(here a repeating 64 KiB pattern per buffer: what a receiving socket or a DMA-capable device layer "
                                            "leaves behind), no memcpy between producer and H2D"}
            if a.workload in ("c2", "ns4096"):
                # BASELINE configs[4]'s per-GPU share end to end: 4096 NBFM streams, /64 + FIR9 + deemph + arbitrary_resample
                from rtlsdr_amd.capi import RESAMPLE_ARBITRARY, load
                c3 = RtlfmCfg.default(downsample=64, downsample_passes=6, comp_fir_size=9, rate_out=16000, deemph=1,
                                      deemph_a=load().rtlfm_deemph_a(16000, 75), rate_out2=22050, resampler=RESAMPLE_ARBITRARY,
                                      block_len=L, max_blocks=1)
                for mode in ("push", "acquire"):
                    r3 = run_native(c3, 4096, mode, 2.0)
                    if r3:
                        out.setdefault("c3_4096_streams", {})[mode] = {
                            "GB/s_in": r3["GB/s_in"], "value": r3["Msamples/s"], "frac_of_pinned_h2d": round(r3["GB/s_in"] / h2d, 3)}
                if "c3_4096_streams" in out:
                    out["c3_4096_streams"]["what"] = (f"configs[4]'s share of one GPU end to end: 4096 streams x 1 buffer x {L} B per run, "
                                                      "u8 IQ in host memory -> int16 audio at 22.05 kHz in host memory")
            return out
    host = job.iq[:S, :L].contiguous().cpu().numpy()  # pageable, as a driver's transfer buffers are
    nthreads = int(os.environ.get("RTLFM_E2E_THREADS", "16"))
    with GpuDemod(cfg, S, local_rank) as g, ThreadPoolExecutor(max_workers=nthreads) as pool:
        lib, hnd = g.lib, g._h
        cap = lib.rtlfm_result_cap(C.byref(cfg)) + 16
        out = np.empty((S, cap), dtype=np.int16)
        lens = np.zeros(S, dtype=np.int32)

        def push_range(t):
            for s in range(t, S, nthreads):
                r = lib.rtlfm_gpu_push(hnd, s, host[s].ctypes.data, L)
                assert r == 0, r

        def push_all():
            list(pool.map(push_range, range(nthreads)))

        push_all(); g.full_demod(); push_all()
        lib.rtlfm_gpu_fetch_all(hnd, out.ctypes.data, cap, lens.ctypes.data)  # warm: ring, mirrors, clocks
        runs, t0 = 0, time.perf_counter()
        while True:
            g.full_demod()   # run k in flight ...
            push_all()       # ... callbacks fill the other half
            r = lib.rtlfm_gpu_fetch_all(hnd, out.ctypes.data, cap, lens.ctypes.data)
            assert r == 0, r
            runs += 1
            if time.perf_counter() - t0 > seconds and runs >= 3:
                break
        dt = time.perf_counter() - t0
    return {"value": round(runs * S * (L // 2) / dt / 1e6, 1), "unit": "Msamples/s",
            "GB/s_in": round(runs * nbytes / dt / 1e9, 2), "pinned_h2d_GB/s": round(h2d_ceiling(), 1), "harness": "python",
            "what": f"{S} streams x 1 buffer x {L} B per run, {runs} pipelined runs in {dt:.2f} s: {nthreads} threads rtlfm_gpu_push "
                    f"(pageable -> pinned ring) | rtlfm_gpu_run (async H2D + kernels) | rtlfm_gpu_fetch_all; "
                    f"bounded by the host memcpy into the ring and PCIe, not by the kernels"}


class ApartRows:
    """int16 [rows, cols] device memory for a kernel's OUTPUT, a quarter of the HBM away from the input it is
    computed from (rtlfm_gpu_malloc_apart: the read and the write stream of a launch get in each other's way when
    they share a 72 GB quarter of the MI355X's memory - DESIGN.md section 3).  Quacks like the torch tensor it replaces."""

    def __init__(self, rows, cols, other_ptr, other_bytes, device, budget_gb=16):
        import ctypes as C
        from rtlsdr_amd.capi import check, load
        self.lib = load()
        self.rows, self.cols = rows, cols
        p, apart, ms, walked = C.c_void_p(), C.c_int(), C.c_double(), C.c_size_t()
        check(self.lib.rtlfm_gpu_malloc_apart_ex(device, rows * cols * 2, other_ptr, other_bytes, budget_gb << 30, C.byref(p), C.byref(apart),
                                                 C.byref(ms), C.byref(walked)), "rtlfm_gpu_malloc_apart_ex")
        self.ptr, self.apart = p.value, bool(apart.value)
        self.search_ms, self.walked_mb = round(ms.value, 1), walked.value >> 20  # what finding the placement cost

    def data_ptr(self):
        return self.ptr

    def stride(self, dim):
        return self.cols if dim == 0 else 1

    def free(self):
        if self.ptr:
            self.lib.rtlfm_gpu_free(self.ptr)
            self.ptr = None


class _DevMem:
    """A raw device pointer dressed for torch.as_tensor (__cuda_array_interface__, which the ROCm build reads too)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class PlacedPair(ApartRows):
    """Input AND output of a launch placed together (rtlfm_gpu_place_pair): `gen` (a torch uint8 [S, n] tensor) is copied into
    the placed input, which is what `.iq` then is - a torch view of library-owned memory; the rows quack like ApartRows."""

    def __init__(self, rows, cols, gen, device, budget_gb=64, max_tries=3):
        import ctypes as C

        import torch
        from rtlsdr_amd.capi import check, load
        self.lib = load()
        self.rows, self.cols = rows, cols
        pin, pout, apart, tries, ms, walked = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int(), C.c_double(), C.c_size_t()
        check(self.lib.rtlfm_gpu_place_pair(device, gen.numel(), rows * cols * 2, budget_gb << 30, max_tries, C.byref(pin), C.byref(pout),
                                            C.byref(apart), C.byref(tries), C.byref(ms), C.byref(walked)), "rtlfm_gpu_place_pair")
        self.ptr, self.in_ptr, self.apart = pout.value, pin.value, bool(apart.value)
        self.search_ms, self.walked_mb, self.tries = round(ms.value, 1), walked.value >> 20, tries.value
        self.iq = torch.as_tensor(_DevMem(self.in_ptr, gen.shape, "|u1"), device=gen.device)
        assert self.iq.data_ptr() == self.in_ptr
        self.iq.copy_(gen)
        torch.cuda.synchronize()

    def free(self):
        super().free()
        self.iq = None
        if self.in_ptr:
            self.lib.rtlfm_gpu_free(self.in_ptr)
            self.in_ptr = None


def ceiling_leg(job, local_rank):
    """This box's own HBM ceilings (nominal 8 TB/s AND a measured ceiling), same invocation:
    the front end's access pattern without its arithmetic (rtlfm_gpu_bw_probe, bw_probe_kernel.h) over
    4 GiB - read only, and read + write at the workload's byte ratio (the PCM is 1/16 of the input bytes
    at /16: one byte stored per 16 read)."""
    import ctypes as C
    from rtlsdr_amd.capi import load
    lib = load()
    wd = max(1, int(round(2.0 / max(job.alg_bytes_per_sample - 2.0, 2.0 / 64))))
    rd, rw, rwc, wf = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    r = lib.rtlfm_gpu_bw_probe(local_rank, 4 << 30, wd, 20, C.byref(rd), C.byref(rw), C.byref(rwc), C.byref(wf))
    if r < 0:
        return None
    return {"read_only": round(rd.value, 1), "read_write": round(rw.value, 1), "read_write_colocated": round(rwc.value, 1),
            "apart_found": bool(r), "unit": "GB/s",
            "write_fraction": round(wf.value, 5), "workload_write_fraction": round((job.alg_bytes_per_sample - 2.0) / 2.0, 5),
            "how": "rtlfm_gpu_bw_probe: 8192 waves x 8 KiB tiles, non-temporal coalesced dwordx4 loads with the next "
                   "tile in flight, 4 waves/SIMD, 4 GiB, 20 launches each, HIP events; read_write = (read + written bytes) / time "
                   "with the written bytes in another 72 GB quarter of the HBM than the read ones (as bench.py places its own "
                   "output: config.output_apart), read_write_colocated = both inside one allocation"}


def _timed(job, warm, K):
    """warm untimed steps, then K timed ones: (HIP-event ms per front-end launch, wall ms per step)."""
    for _ in range(warm):
        job.step()
    job.sync()
    job.g.timing_enable(True); job.g.timing_read()
    t0 = time.perf_counter()
    for _ in range(K):
        job.step()
    job.sync()
    dt = time.perf_counter() - t0
    ms, cnt = job.g.timing_read()
    job.g.timing_enable(False)
    return ms / max(cnt, 1), dt / K * 1e3


class ColocatedRows:
    """int16 [rows, cols] device memory that SHARES its class of the HBM with its input - what a caller who knows nothing of
    the placement often gets: ONE allocation holds a copy of the input and, behind it, the output rows (offsets inside an
    allocation do not change the class).  `colocated`: what rtlfm_gpu_placement_probe says about the two."""

    def __init__(self, rows, cols, iq, dev):
        import ctypes as C
        import torch
        from rtlsdr_amd.capi import load
        self.lib = load()
        self.rows, self.cols = rows, cols
        nin = (iq.numel() + 255) & ~255
        self.buf = torch.empty(nin + rows * cols * 2, dtype=torch.uint8, device=dev)
        self.buf[:iq.numel()].view(iq.shape).copy_(iq)
        torch.cuda.synchronize()
        self.iq_ptr = self.buf.data_ptr()
        self.ptr = self.iq_ptr + nin
        rd, rw = C.c_double(), C.c_double()
        r = self.lib.rtlfm_gpu_placement_probe(dev.index or 0, self.iq_ptr, iq.numel(), self.ptr, rows * cols * 2, C.byref(rd), C.byref(rw))
        self.colocated = r == 0
        self.apart = not self.colocated

    def data_ptr(self):
        return self.ptr

    def stride(self, dim):
        return self.cols if dim == 0 else 1

    def free(self):
        self.buf = None
        self.ptr = None


def _leg_entry(name, job, launch_ms, step_ms, K, ceiling):
    alg = job.alg_bytes_per_sample * job.samples
    ach = alg / (launch_ms * 1e-3) / 1e9
    e = {"workload": f"{name}: " + job.describe(), "kernel": job.kernel_name(), "steps": K,
         "launch_ms": round(launch_ms, 4), "ms_per_step": round(step_ms, 4),
         "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
         # the whole step (front end + whatever follows it on the GPU: the audio tail), wall clock
         "step_frac": round(alg / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
         "value": round(job.samples / (step_ms * 1e-3) / 1e6, 1), "unit": "Msamples/s",
         "algorithmic_bytes_per_sample": round(job.alg_bytes_per_sample, 5)}
    if getattr(job, "output_apart", None) is not None:
        e["output_apart"] = job.output_apart
    if ceiling and job.a.tail != "power":
        e["frac_of_ceiling"] = round(ach / ceiling["read_write"], 4)
    if job.a.tail != "power" and hasattr(job.g, "get_option"):
        # where the handle put its OWN buffers (the audio tail's work buffers, the emit-mode buffer): 1 = apart from the
        # input, 0 = searched and not found, -1 = this configuration has none
        e["handle_placement"] = {k: job.g.get_option(k) for k in ("res_apart", "deep_apart", "placement_ms", "placement_walked_mb")}
    return e


def also_legs(a, job, dev, local_rank, rank, ceiling, valu_insts=None):
    """Every other BASELINE shape, timed by this invocation on this box (so that each has a driver-run figure):
    `ns4096x1` and `c2` re-read the default workload's resident 4 GiB (4096 streams x ONE buffer per launch - what
    a live capture hands over per callback round, src/rtl_fm.c:1339-1343 - and configs[1]'s 256 streams x 64
    buffers); `c3`, `c1`, `wbfm`, `c4` get inputs of their own.  100 untimed + --also-steps timed launches each."""
    import torch
    from rtlsdr_amd.capi import RtlfmCfg
    from rtlsdr_amd.demod import GpuDemod
    out = {}
    L, K = a.block_len, a.also_steps
    total = job.iq.numel()

    class Reuse:
        """the resident bytes through the default chain under another (streams, buffers) shape"""

        def __init__(self, S, nb, per_stream, block_len=None, colocated=False, two_outputs=False):
            self.a = a
            self.S, self.nb, self.per_stream = S, nb, per_stream
            self.L = block_len or L
            cfg = RtlfmCfg.from_buffer_copy(bytes(job.cfg))
            cfg.max_blocks = nb
            cfg.block_len = self.L
            self.g = GpuDemod(cfg, S, local_rank)
            cap = self.g.result_cap(nb)
            have = job.out.rows * job.out.cols if isinstance(job.out, ApartRows) else 0
            self.iq_ptr = job.iq.data_ptr()
            if colocated:
                # input and output in ONE allocation: the same class of the HBM by construction (offsets inside an allocation
                # do not matter: LAB.md II 3.1) - a copy of the input, the output rows behind it
                self.o = ColocatedRows(S, cap, job.iq, dev)
                self.iq_ptr = self.o.iq_ptr
            elif S * cap <= have:
                # the default workload's own output buffer, whose placement is known, under this leg's row length
                outer = job.out

                class View:
                    ptr, apart = outer.ptr, outer.apart

                    def data_ptr(self):
                        return self.ptr

                    def stride(self, dim):
                        return cap if dim == 0 else 1

                    def free(self):
                        pass
                self.o = View()
            else:
                self.o = ApartRows(S, cap, job.iq.data_ptr(), job.iq.numel(), local_rank, 64)
            self.output_apart = self.o.apart
            # two_outputs: the launches alternate between two output buffers - what a consumer that double-buffers sees.  ONE
            # 256 MiB output rewritten by every launch stays in the 256 MiB Infinity Cache from launch to launch (LAB.md I.12)
            self.o2 = ApartRows(S, cap, job.iq.data_ptr(), job.iq.numel(), local_rank, 64) if two_outputs else None
            if self.o2 is not None:
                self.output_apart = bool(self.o.apart and self.o2.apart)
            self.flip = 0
            self.n = torch.zeros(S, dtype=torch.int32, device=dev)
            self.samples = S * nb * self.L // 2
            self.alg_bytes_per_sample = job.alg_bytes_per_sample

        def step(self):
            o = self.o
            if self.o2 is not None:
                self.flip ^= 1
                o = self.o2 if self.flip else self.o
            self.g.run_device(self.iq_ptr, self.per_stream, self.nb, o.data_ptr(), o.stride(0), self.n.data_ptr())

        def sync(self):
            self.g.sync()

        def describe(self):
            return (f"{self.S} streams/GPU x {self.nb} buffer(s) x {self.L} B per launch, the default workload's chain and bytes"
                    + (", two output buffers in turn" if self.o2 is not None else ""))

        def kernel_name(self):
            return job.kernel_name()

        def close(self):
            self.g.close(); self.o.free()
            if self.o2 is not None:
                self.o2.free()

    if a.workload == "ns4096" and not a.boxcar and total == 4096 * 4 * L:
        # ns4096_colocated: the headline launch with its output where input and output SHARE a class of the HBM (what a
        # caller gets who allocates one after the other): the other end of the placement's effect;
        # c2_16k: configs[1] on the reference's own 16384-byte buffers (src/rtl_fm.c:1605)
        for name, S, nb, per, bl, colo in (("ns4096x1", 4096, 1, total // 4096, L, False), ("c2", 256, 64, total // 256, L, False),
                                           ("ns4096_colocated", 4096, 4, total // 4096, L, True),
                                           ("ns4096_two_outputs", 4096, 4, total // 4096, L, False),
                                           ("c2_16k", 256, total // 256 // 16384, total // 256, 16384, False)):
            r = Reuse(S, nb, per, bl, colo, two_outputs=name == "ns4096_two_outputs")
            launch_ms, step_ms = _timed(r, 400 if nb == 1 else 100, 4 * K if nb == 1 else K)
            e = _leg_entry(name, r, launch_ms, step_ms, 4 * K if nb == 1 else K, ceiling)
            r.g.clock_probe(True); r.step(); st = r.g.clock_stamps(); r.g.clock_probe(False); r.sync()
            if st is not None:
                e["waves_per_stream"] = len(st) // S
            if colo:
                e["colocated"] = r.o.colocated
                e["how"] = "a copy of the input and the output rows inside ONE allocation"
            out[name] = e
            r.close()
    for name in ("c3", "c1", "wbfm", "scanner", "c4"):
        b = workload_args(a, name)
        b.crowded = True  # this process holds the default workload's buffers (and what the legs before left behind)
        j = None
        try:
            j = (PowerJob if b.tail == "power" else FmJob)(b, dev, local_rank, rank)
            launch_ms, step_ms = _timed(j, 100, K)
            e = _leg_entry(name, j, launch_ms, step_ms, K, ceiling)
            if b.tail == "power":
                e["valu_issue"] = j.valu_issue(launch_ms, valu_insts)
            else:
                e["output_placement"] = j.placement
            out[name] = e
        except Exception as ex:  # noqa: BLE001 - a leg that fails is reported, not fatal
            out[name] = {"error": repr(ex)}
        if j is not None:
            j.close()
        del j
        torch.cuda.empty_cache()
    return out


def time_scatter(dist, rank, world, dev, streams_per_rank, bytes_per_stream, reps=3):
    """The one optional exchange of the path: all IQ of a step lands on rank 0's
    GPU and every rank receives its contiguous stream range (shard.scatter_streams: isend/recv
    per peer, one xGMI link each under RCCL).  Bounded: at most 1 GiB per peer."""
    import torch
    from rtlsdr_amd import shard
    per = min(streams_per_rank, max(1, (1 << 30) // bytes_per_stream))
    total = per * world
    gloo = dist.get_backend() != "nccl"
    where = "cpu" if gloo else dev
    root = torch.empty((total, bytes_per_stream), dtype=torch.uint8, device=where) if rank == 0 else None
    if root is not None:
        root.random_(0, 256)
    ms = []
    for i in range(reps + 1):
        dist.barrier()
        if not gloo:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        mine = shard.scatter_streams(root, total, bytes_per_stream, device=where)
        if not gloo:
            torch.cuda.synchronize()
        dist.barrier()
        if i:
            ms.append((time.perf_counter() - t0) * 1e3)
        assert mine.shape[0] == per
    t = torch.tensor([min(ms)], dtype=torch.float64, device=where)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sent = per * (world - 1) * bytes_per_stream
    return {"ms": round(float(t.item()), 3), "bytes_from_root": sent,
            "GB/s_from_root": round(sent / (float(t.item()) * 1e-3) / 1e9, 1),
            "what": f"root -> {world - 1} peers, {per} streams x {bytes_per_stream} B each, {dist.get_backend()}"}


# ------------------------------------------------------------------- workloads ----

class FmJob:
    """rtl_fm's chain on S streams x NB buffers per step."""

    def __init__(self, a, dev, local_rank, rank):
        import torch
        from rtlsdr_amd import synth
        from rtlsdr_amd.capi import (ATAN_FAST, ATAN_LUT, ATAN_STD, RESAMPLE_ARBITRARY, RESAMPLE_LOW_PASS_REAL,
                                     RtlfmCfg, load)
        from rtlsdr_amd.demod import GpuDemod
        self.a, self.torch = a, torch
        lib = load()
        atan = {"std": ATAN_STD, "fast": ATAN_FAST, "lut": ATAN_LUT}[a.atan]
        D = a.boxcar if a.boxcar else 1 << a.passes
        self.D = D
        rate_out = int(a.fs / D)
        kw = dict(downsample=D, downsample_passes=a.passes, comp_fir_size=9 if a.fir9 else 0, custom_atan=atan,
                  rate_out=rate_out, block_len=a.block_len, max_blocks=a.blocks, dc_block_raw=1 if a.rdc else 0,
                  squelch_level=int(getattr(a, "squelch", 0)))
        self.out_ratio = 1.0
        if a.tail == "c3":
            kw.update(rate_out=16000, deemph=1, deemph_a=lib.rtlfm_deemph_a(16000, 75), rate_out2=22050,
                      resampler=RESAMPLE_ARBITRARY)
            self.out_ratio = 176.0 / 128.0  # per 262144-B buffer: 2048 -> 2822
        elif a.tail == "wbfm":
            kw.update(rate_out=170000, deemph=1, deemph_a=lib.rtlfm_deemph_a(170000, 75), rate_out2=32000,
                      resampler=RESAMPLE_LOW_PASS_REAL)
            self.out_ratio = 32000.0 / 170000.0
        self.cfg = RtlfmCfg.default(**kw)
        S, NB, L = a.streams, a.blocks, a.block_len
        self.samples = S * NB * L // 2
        amp = 40.0 if a.atan == "fast" else 60.0  # -A fast overflows above |z| ~ 724
        if a.atan == "fast" and a.boxcar:
            amp = min(40.0, 500.0 / a.boxcar)
        if a.pmc_child:
            # counter passes only weigh bytes: one fill kernel instead of the signal generator's thousands of
            # small dispatches (rocprofv3's counter mode stalled on those at 4096 streams)
            self.iq = torch.randint(0, 256, (S, NB * L), dtype=torch.uint8, device=dev)
        else:
            self.iq = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=a.fs, dev_hz=75e3 if a.fs > 2e6 or a.tail == "wbfm" else 5e3,
                                           amplitude=amp, first_stream=rank * S)
        self.g = GpuDemod(self.cfg, S, local_rank, options=({"apart_budget_gb": 64} if getattr(a, "crowded", False) else None))
        self.g.set_path(a.path)
        cap = self.g.result_cap(NB) + int(os.environ.get("RTLFM_BENCH_ROW_PAD", "0"))  # experiments: rows off the 128-byte lines
        # the output a quarter of the HBM away from the input (data layout); --colocate 1 = wherever torch puts it
        self._parked = []
        if a.colocate:
            self.out = torch.empty((S, cap), dtype=torch.int16, device=dev)
        else:
            # "apart" is a minority of the memory and where it lies differs from box to box: if the
            # search comes back empty-handed, the input moves (a second copy somewhere else; the first stays parked so that
            # the allocator cannot hand the same place out again) and the search runs once more - three tries in all
            # (r04: a first search walked 152 GB in 4.7 s and found nothing; with the input moved the first candidate was apart.
            # Round 5: the search itself is bounded - eight candidates of different sizes, 16 GiB held at most)
            # Round 6: the harness owns BOTH sides, so it asks for the pair in one call (rtlfm_gpu_place_pair: the library's own
            # retry - park the input, take another, search again - in exported form; up to three searches, 64 GiB each here,
            # where the library's default for a handle's own buffers stays 16 GiB).  The generated input moves into the placed
            # memory once; run_device only ever sees pointers.
            self.out = PlacedPair(S, cap, self.iq, local_rank, budget_gb=64, max_tries=1 if a.pmc_child else 3)
            self.iq = self.out.iq
            tries = self.out.tries
        self.output_apart = bool(getattr(self.out, "apart", False))
        self.placement = {"search_ms": getattr(self.out, "search_ms", None), "walked_mb": getattr(self.out, "walked_mb", None)}
        if not a.colocate:
            self.placement.update(searches=tries, budget_gb=64, how="rtlfm_gpu_place_pair: input and output chosen together")
        self.out_len = torch.zeros(S, dtype=torch.int32, device=dev)
        self.local_rank = local_rank
        # u8 I + u8 Q in, int16 PCM out at 1/D (x the resampling ratio)
        self.alg_bytes_per_sample = 2.0 + 2.0 / D * self.out_ratio

    def step(self):
        self.g.run_device(self.iq.data_ptr(), self.iq.stride(0), self.a.blocks, self.out.data_ptr(), self.out.stride(0),
                          self.out_len.data_ptr())

    def sync(self):
        self.g.sync()

    def gate(self):
        from rtlsdr_amd.capi import RtlfmCfg
        from rtlsdr_amd.demod import GpuDemod
        a = self.a
        cs, cb = min(a.streams, 8), min(a.blocks, 2)
        ccfg = RtlfmCfg.from_buffer_copy(bytes(self.cfg))
        ccfg.max_blocks = cb
        sub = self.iq[:cs, :cb * a.block_len].contiguous()
        with GpuDemod(ccfg, cs, self.local_rank) as gc:
            gc.set_path(a.path)
            o, n = gc.run_torch(sub)
            gc.sync()
        return (ccfg, sub.cpu().numpy(), o.cpu().numpy(), n.cpu().numpy())

    def cpu_baseline(self, seconds, gate):
        cs = min(self.a.streams, os.cpu_count() or 1)
        sample = self.iq[:cs, :min(self.a.blocks, 2) * self.a.block_len].contiguous().cpu().numpy()
        return cpu_baseline(self.cfg, sample, seconds, gate)

    def describe(self):
        a, D = self.a, self.D
        front = f"low_pass boxcar /{D}" if a.boxcar else f"{a.passes}x fifth_order (/{D})" + (" + FIR9" if a.fir9 else "")
        tail = {"c3": " + deemph + arbitrary_resample 16k -> 22050", "wbfm": " + deemph + low_pass_real 170k -> 32k"}.get(a.tail, "")
        front = ("dc_block_raw_filter + " if a.rdc else "") + front + (f" + squelch {a.squelch}" if getattr(a, "squelch", 0) else "")
        return (f"rtl_fm -A {a.atan}: {a.streams} streams/GPU x {a.blocks} buffers x {a.block_len} B u8 IQ @{a.fs / 1e6:g} MS/s, "
                f"{front} + polar discriminant{tail} -> int16 PCM")

    def kernel_name(self):
        return ("k_boxcar_scan (convert+rotate+low_pass+discriminant)" if self.a.boxcar
                else "k_fused (convert+rotate+fifth_order[+fir9]+discriminant)")

    def close(self):
        self.g.close()
        if hasattr(self.out, "free"):
            if isinstance(self.out, PlacedPair):
                self.iq = None  # a view of memory the pair owns
            self.out.free()
        self._parked = []


class PowerJob:
    """rtl_power's scanner() on S tuning states x NB reads per step (config 4)."""

    def __init__(self, a, dev, local_rank, rank):
        import torch
        from rtlsdr_amd import synth
        from rtlsdr_amd.capi import RtlpowerCfg
        from rtlsdr_amd.power import GpuPower
        self.a, self.torch = a, torch
        self.bin_e = 14
        L = a.block_len
        assert L >= (2 << self.bin_e), "config 4 reads hold one 16384-point frame"
        self.cfg = RtlpowerCfg.default(bin_e=self.bin_e, window=1, buf_len=L)
        S, NB = a.streams, a.blocks
        self.samples = S * NB * L // 2
        if a.pmc_child:
            self.iq = torch.randint(0, 256, (S, NB * L), dtype=torch.uint8, device=dev)
        else:
            self.iq = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=a.fs, dev_hz=50e3, first_stream=rank * S)
        self.g = GpuPower(self.cfg, S, local_rank)
        self.local_rank = local_rank
        # 2 B read per complex sample + the int64 accumulators written back once per launch
        self.alg_bytes_per_sample = 2.0 + 8.0 * S * (1 << self.bin_e) / self.samples
        self.D = 1

    def step(self):
        self.g.scan_device(self.iq.data_ptr(), self.iq.stride(0), self.a.blocks)

    def sync(self):
        self.g.sync()

    def gate(self):
        import numpy as np
        from rtlsdr_amd.power import GpuPower
        cs = min(self.a.streams, 4)
        sub = self.iq[:cs, :2 * self.a.block_len].contiguous()
        with GpuPower(self.cfg, cs, self.local_rank) as gp:
            gp.scan_torch(sub)
            gp.sync()
            got = [gp.fetch(s) for s in range(cs)]
        return (sub.cpu().numpy(), np.stack([g[0] for g in got]), np.array([g[1] for g in got], dtype=np.int32))

    def cpu_baseline(self, seconds, gate):
        cs = min(self.a.streams, os.cpu_count() or 1)
        sample = self.iq[:cs, :2 * self.a.block_len].contiguous().cpu().numpy()
        return cpu_baseline_power(self.cfg, sample, seconds, gate)

    def valu_issue(self, launch_ms, valu_insts):
        """The launch time the kernel's integer VALU instruction count alone takes: SQ_INSTS_VALU of a launch (this
        invocation's own rocprofv3 --pmc child; wave-instructions) x 4 cycles of issue each, over 1024 SIMDs at the
        shader clock the kernel's workgroups measured themselves (rtlpower_gpu_clock_probe)."""
        self.g.clock_probe(True)
        for _ in range(3):
            self.step()
        clk = self.g.clock_read()
        self.g.clock_probe(False)
        out = {"shader_mhz": round(clk[0], 0) if clk else None, "valu_insts_per_launch": valu_insts,
               "how": "SQ_INSTS_VALU per launch from this invocation's rocprofv3 --pmc child run; shader clock from the kernel's own "
                      "s_memtime / s_memrealtime stamps; floor = insts x 4 cycles / (1024 SIMDs x clock)"}
        if valu_insts:
            out["lane_ops_per_sample"] = round(valu_insts * 64.0 / self.samples, 2)
            if clk:
                floor_ms = valu_insts * 4.0 / (1024 * clk[0] * 1e6) * 1e3
                out["floor_ms"] = round(floor_ms, 3)
                out["frac"] = round(floor_ms / launch_ms, 3)
        # ... and against a count made by hand of what a bit-exact radix-2 fix_fft NEEDS per complex sample (the
        # figure above is the kernel's own instruction count and says how busy the VALU is, not how few instructions would do).
        # A butterfly is 4 FIX_MPY - each rounds by itself, so four 16 x 16 multiply-adds whose high halves are the results
        # (v_mad_i32_i16) - 2 v_perm to gather the four high halves into two packed pairs, one packed multiply-add for
        # (m1 - m2, m3 + m4), a >> 1 and the packed sum and difference: 10, two points each; the first two stages' twiddles
        # are real or imaginary (two products less: 7.5 on average); src/rtl_power.c:271-327.  In front: unpack, - 127 - DC,
        # window (int16 wrap), bit-reversed address: 6 per point (:666-668, 581-596, 697-706); behind: re^2 + im^2 and a
        # 64-bit add: 3 (:708-716).  LDS reads and writes, waits and scalar work are not VALU instructions.
        e = self.bin_e
        hand = 6.0 + (10.0 * (e - 2) + 7.5 * 2) / 2.0 + 3.0
        out["lane_ops_hand_count"] = round(hand, 1)
        out["hand_count_what"] = "6 (convert, DC, window, bit-reversed address) + 5 per radix-2 stage (3.75 for the first two) + 3 (|X|^2, int64 add)"
        if clk:
            hand_ms = hand * self.samples / 64.0 * 4.0 / (1024 * clk[0] * 1e6) * 1e3
            out["hand_count_floor_ms"] = round(hand_ms, 3)
            out["frac_of_hand_count"] = round(hand_ms / launch_ms, 3)
        return out

    def describe(self):
        a = self.a
        return (f"rtl_power -w hamming: {a.streams} streams/GPU x {a.blocks} reads x {a.block_len} B u8 IQ @{a.fs / 1e6:g} MS/s, "
                f"2^{self.bin_e}-bin fix_fft + |X|^2 integrate -> int64 avg[]")

    def kernel_name(self):
        return "k_power_scan_big<14> (convert+remove_dc+window+fix_fft+integrate)"

    def close(self):
        self.g.close()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # Before this process touches the GPU: build (hipcc / gcc children) and the PMC child runs.
    traffic, c4_valu = None, None
    under_profiler = any("ROCPROF" in k for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if under_profiler:
        # rocprofv3's preloaded library has initialised the GPU in this process already: nothing here may
        # start compilers, shells or other children that exec onward (the pool forbids that hop), and the
        # PCIe leg's child process would inherit the profiler.  Stale artefacts are an error, not a build.
        a.e2e = 0
        if not a.pmc_child:
            from rtlsdr_amd import build as b
            if b.needs_build():
                print("bench.py: librtlfm_hip.so is stale and this process runs under rocprofv3: build first "
                      "(python -c 'import __graft_entry__ as g; g.build()'), then profile", file=sys.stderr)
                sys.exit(3)
    elif not a.pmc_child:
        import __graft_entry__ as ge
        if rank == 0:
            ge.build()
        want_pmc = a.pmc == 1 or (a.pmc < 0 and world == 1)
        if want_pmc and rank == 0 and world == 1:
            traffic = pmc_traffic(a)
            # rtl_power's bound is VALU issue: its instruction count comes from this invocation too
            if a.tail == "power" or (a.workload == "ns4096" and a.also and not a.boxcar and not a.rdc):
                v = pmc_counters(a if a.tail == "power" else workload_args(a, "c4"), ("SQ_INSTS_VALU",))
                c4_valu = v["SQ_INSTS_VALU"] if v else None

    import torch
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        # RCCL over xGMI on a real node; RTLFM_BENCH_BACKEND=gloo only exists to exercise this
        # control flow where several ranks have to share one GPU (RCCL refuses that)
        backend = os.environ.get("RTLFM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        dist.barrier()  # rank 0 has built the library

    job = (PowerJob if a.tail == "power" else FmJob)(a, dev, local_rank, rank)
    step = job.step

    def fence():
        job.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    # a slice through the product before any timing (rank 0): the same handle type, same kernels;
    # the cpu_baseline leg below compares it with the CPU path's output for the same bytes
    gate = None
    if a.check and rank == 0 and not a.no_cpu_baseline and world == 1:
        gate = job.gate()

    # Untimed: after idle the GPU needs ~50 launches (~50 ms) to reach its steady clock - on most boxes; round 5's last
    # driver-shaped run met one where 20 timed steps behind 100 untimed ones still read 2.4 % under the 2-second sustained
    # figure of the same invocation (0.7225 against 0.7405).  If the caller asks for fewer warm-up steps than 400 (0.3 s), the
    # difference is run first and reported as config.prewarm_steps, so that the K timed steps measure the steady state.
    prewarm = 0 if a.pmc_child else max(0, 400 - a.warmup)
    # ... and what the K steps measure behind EXACTLY --warmup untimed ones (`cold`), first: the figure a caller who warms up
    # as the command line says would see; the steady-state figure below stays `value` (say both)
    cold = None
    if prewarm and not a.pmc_child:
        for _ in range(a.warmup):
            step()
        fence()
        job.g.timing_enable(True)
        job.g.timing_read()
        tc = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        cold_elapsed = time.perf_counter() - tc
        c_ms, c_n = job.g.timing_read()
        job.g.timing_enable(False)
        cold = (cold_elapsed, c_ms / max(c_n, 1))
    for _ in range(prewarm + a.warmup):
        step()
    fence()
    job.g.timing_enable(True)
    job.g.timing_read()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    front_ms, launches = job.g.timing_read()
    path_used = getattr(job.g, "last_path", 2)

    # the sustained leg: the same step for >= a.sustain seconds (HIP events per launch, wall clock over all)
    sustained = None
    if a.sustain > 0:
        n_s = max(a.steps, int(a.sustain / max(elapsed / a.steps, 1e-6)) + 1)
        job.g.timing_read()
        t1 = time.perf_counter()
        for _ in range(n_s):
            step()
        job.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        s_ms, s_n = job.g.timing_read()
        sustained = (n_s, dt, s_ms / max(s_n, 1))
    job.g.timing_enable(False)

    # the in-kernel clock stamps of one more launch, right behind the sustained leg (fused fifth_order path)
    clock = None
    if hasattr(job.g, "clock_probe") and not a.pmc_child:
        job.g.clock_probe(True)
        for _ in range(3):
            step()
        clock = job.g.clock_read()
        job.g.clock_probe(False)

    # the optional legs report; they must never cost the contract line
    def guarded(what, fn, *args, **kw):
        try:
            return fn(*args, **kw)
        except Exception as ex:  # noqa: BLE001
            print(f"bench.py: the {what} leg failed: {ex!r}", file=sys.stderr)
            return {"error": repr(ex)}

    ceiling, also = None, None
    if rank == 0 and world == 1 and a.tail != "power" and not a.pmc_child and a.ceiling:
        ceiling = guarded("ceiling", ceiling_leg, job, local_rank)
        if ceiling and "error" in ceiling:
            ceiling = None
        if a.workload == "ns4096" and a.also and not a.boxcar and not a.rdc:
            also = guarded("also", also_legs, a, job, dev, local_rank, rank, ceiling, c4_valu)

    e2e = None
    if a.e2e and rank == 0 and world == 1 and a.tail != "power" and not a.pmc_child:
        e2e = guarded("e2e", e2e_leg, a, job, local_rank)
    # N > 1: at eight GPUs the curve is the HOST side (8 x ~56 GB/s of pinned H2D through one box's memory), so every
    # rank times its own PCIe-inclusive loop at the same moment (all ranks between two barriers) and reports it with
    # its pinned-H2D rate and whether its output got its placement
    e2e_rank = None
    if a.e2e and world > 1 and a.tail != "power" and not a.pmc_child:
        dist.barrier()
        try:
            r_ = e2e_leg(a, job, local_rank, seconds=2.0, quick=True)
            e2e_rank = (float(r_["GB/s_in"]), float(r_["pinned_h2d_GB/s"]))
        except Exception as ex:  # noqa: BLE001 - the leg is a report, not the benchmark
            print(f"bench.py: rank {rank}: e2e leg failed: {ex!r}", file=sys.stderr)
            e2e_rank = (0.0, 0.0)
        dist.barrier()

    n_devices, scatter, per_rank = 1, None, None
    if dist:
        cdev = dev if dist.get_backend() == "nccl" else "cpu"
        # every rank's own clock and front-end launch time, so that host-side contention shows
        mine = torch.tensor([elapsed / a.steps * 1e3, front_ms / max(launches, 1),
                             1.0 if getattr(job, "output_apart", False) else 0.0,
                             e2e_rank[0] if e2e_rank else -1.0, e2e_rank[1] if e2e_rank else -1.0], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [round(float(x[0]), 4) for x in allr], "launch_ms": [round(float(x[1]), 4) for x in allr],
                    "output_apart": [bool(x[2] > 0.5) for x in allr],
                    "backend": dist.get_backend(), "world_size": dist.get_world_size()}
        if e2e_rank:
            per_rank["e2e_GB/s_in"] = [round(float(x[3]), 2) for x in allr]
            per_rank["pinned_h2d_GB/s"] = [round(float(x[4]), 1) for x in allr]
            per_rank["e2e_GB/s_in_sum"] = round(sum(float(x[3]) for x in allr), 1)
            per_rank["e2e_what"] = ("every rank's own push / run / fetch loop (rtlfm_gpu_push from 8 native threads pinned to its device's NUMA "
                                    "node, 1024 streams x 1 buffer per run), all ranks at the same time: PCIe- and host-memory-bound")
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # n_gpus = the devices that actually joined (ranks sharing a device count once)
        seen = torch.zeros(max(1, torch.cuda.device_count()), dtype=torch.int32, device=cdev)
        seen[local_rank] = 1
        dist.all_reduce(seen, op=dist.ReduceOp.MAX)
        n_devices = int(seen.sum().item())
        if a.scatter:
            scatter = time_scatter(dist, rank, world, dev, a.streams, a.blocks * a.block_len)

    if rank == 0:
        alg_bytes = job.alg_bytes_per_sample * job.samples  # per launch of the dominant kernel
        value = world * job.samples * a.steps / elapsed / 1e6
        launch_ms = front_ms / max(launches, 1)
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9 if launches else None
        roof = {
            "bound": "hbm",
            "achieved": round(achieved, 1) if achieved else None,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
            "traffic": traffic["bytes"] if traffic else None,
            "kernel": job.kernel_name(),
            "launch_ms": round(launch_ms, 4),
            "algorithmic_bytes_per_sample": round(job.alg_bytes_per_sample, 5),
            "algorithmic_bytes_per_launch": int(alg_bytes),
        }
        if traffic:
            roof["traffic_detail"] = traffic
            roof["traffic_over_algorithmic"] = round(traffic["bytes"] / alg_bytes, 4)
        if a.tail in ("c3", "wbfm"):
            # the whole step (front end + audio tail kernels), wall clock
            roof["step_frac"] = round(alg_bytes / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS, 4)
        if a.tail == "power":
            roof["note"] = ("not HBM-bound: integer VALU issue in LDS-resident radix-2 stages; valu_issue = the launch time "
                            "the instruction count alone takes at the clock the launch ran at (DESIGN.md section 4.5)")
            roof["valu_issue"] = job.valu_issue(front_ms / max(launches, 1), c4_valu)
        if sustained:
            n_s, dt, s_launch = sustained
            roof["sustained"] = {
                "seconds": round(dt, 2), "steps": n_s, "ms_per_step": round(dt / n_s * 1e3, 4),
                "launch_ms": round(s_launch, 4),
                "achieved": round(alg_bytes / (s_launch * 1e-3) / 1e9, 1),
                "frac": round(alg_bytes / (s_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "value": round(world * job.samples * n_s / dt / 1e6, 1),
            }
        if cold:
            roof["cold"] = {"what": f"the {a.steps} steps timed right behind exactly --warmup = {a.warmup} untimed ones, before the pre-warm "
                                    f"(the GPU's clock is still ramping: ~50 launches from idle)",
                            "ms_per_step": round(cold[0] / a.steps * 1e3, 4), "launch_ms": round(cold[1], 4),
                            "frac": round(alg_bytes / (cold[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "value": round(world * job.samples * a.steps / cold[0] / 1e6, 1)}
        if clock:
            roof["shader_mhz"] = round(clock[0], 0)
            roof["kernel_span_ms"] = round(clock[1], 4)
        if ceiling and achieved:
            roof["ceiling"] = ceiling
            roof["frac_of_ceiling"] = round(achieved / ceiling["read_write"], 4)
            if sustained:
                roof["sustained"]["frac_of_ceiling"] = round(roof["sustained"]["achieved"] / ceiling["read_write"], 4)
        res = {
            "metric": "IQ Msamples/s demodulated (whole node)",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": n_devices,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16/int32 fixed point (fp64 atan2)" if a.tail != "power" else "int16 fixed point (int64 accumulate)",
            "data": "synthetic",
            "config": {
                "workload": f"{a.workload}: " + job.describe(),
                "streams_per_gpu": a.streams, "buffers_per_step": a.blocks, "block_len": a.block_len, "passes": a.passes,
                "path": {1: "staged", 2: "fused"}.get(path_used, str(path_used)),
                "parallelism": f"streams sharded {a.streams}/GPU over {n_devices} GPU(s), {world} rank(s), no data-path collective",
                "ranks": world,
                "requested_gpus": int(os.environ.get("RTLFM_BENCH_REQUESTED_GPUS", a.gpus)),
                "prewarm_steps": prewarm,
                "output_apart": getattr(job, "output_apart", None),
                "output_placement": getattr(job, "placement", None),
                "handle_placement": ({k: job.g.get_option(k) for k in ("res_apart", "deep_apart", "placement_ms", "placement_walked_mb")}
                                     if a.tail != "power" else None),
                "device": torch.cuda.get_device_name(local_rank),
            },
            "roofline": roof,
        }
        # which placement produced the headline, in the open: the searches it took and their budget; the same launch
        # with its output wherever the allocator puts it is also.ns4096_colocated
        pl = getattr(job, "placement", None)
        if pl:
            res["placement"] = {"output_apart": getattr(job, "output_apart", None), "searches": pl.get("searches"), "budget_gb": pl.get("budget_gb"),
                                "search_ms": pl.get("search_ms"), "walked_mb": pl.get("walked_mb"), "library_default_budget_gb": 16,
                                # would a caller with the library's default budget and no input move have got this placement?
                                "within_default_budget": (pl.get("searches") == 1 and (pl.get("walked_mb") or 0) <= 16 * 1024
                                                          and bool(getattr(job, "output_apart", False))),
                                "colocated_frac": (also or {}).get("ns4096_colocated", {}).get("frac") if isinstance(also, dict) else None}
        if also:
            res["also"] = also
        if e2e:
            res["e2e"] = e2e
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = job.cpu_baseline(a.cpu_seconds, gate)
            res["cpu_baseline"]["parity_checked"] = gate is not None
        else:
            res["cpu_baseline"] = None
        if per_rank:
            per_rank["ms_per_step_min"] = min(per_rank["ms_per_step"]); per_rank["ms_per_step_max"] = max(per_rank["ms_per_step"])
            res["per_rank"] = per_rank
        if scatter:
            res["scatter"] = scatter
        print(json.dumps(res))
    job.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
