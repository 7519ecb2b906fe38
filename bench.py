#!/usr/bin/env python3
"""bench.py — IQ Msamples/s demodulated on MI355X, with roofline and CPU baseline.

One "step" is one pass of the hot path (u8 IQ -> rotate -> fifth_order x passes
-> FM discriminant -> int16 PCM, reference src/rtl_fm.c:1326-1338 + 1179-1272)
over one batch of synthetic input that is already resident in HBM.

Default workload = BASELINE.json configs[1]: 256 batched 2.4 MS/s WBFM streams,
`-M fm -s 150k -m 1.3M -F 0` (4 fifth_order passes, /16, polar_discriminant),
64 callback buffers of 262144 B per stream per step (4 GiB of IQ per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Streams are independent: rank r owns its own `--streams` streams (weak
scaling), there is no data-path collective; RCCL is used only for the barrier
and the max-over-ranks of the elapsed time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a launch is ~0.9 ms; after idle the GPU needs ~50 of them to reach its steady clock
    # (30 timed launches behind 5 warm-ups measure the ramp: 0.91 ms against 0.83 ms)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--blocks", type=int, default=64, help="callback buffers per stream per step")
    ap.add_argument("--block-len", type=int, default=262144)
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--boxcar", type=int, default=0,
                    help="D > 0: the reference's default low_pass boxcar /D instead of fifth_order passes "
                         "(config 1 / -M wbfm shaped work; not the default line)")
    ap.add_argument("--fir9", type=int, default=0)
    ap.add_argument("--atan", choices=["std", "fast", "lut"], default="std")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 staged, 2 fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--check", type=int, default=1, help="have the cpu_baseline leg compare a slice of the GPU output with the CPU path")
    ap.add_argument("--scatter", action="store_true",
                    help="N > 1: also time shard.scatter_streams (root GPU -> its owner GPUs) of one step's IQ over RCCL")
    return ap.parse_args()


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` outside torchrun: start one rank per GPU ourselves.

    Runs before anything in this process has touched the GPU: the ranks are children of a
    `python -m torch.distributed.run` subprocess (never a re-exec of a process that holds the
    device).  torch.cuda.device_count() only counts, it does not initialise the runtime here.
    The JSON line is printed by rank 0 of the child job; this process only relays its exit code.
    """
    import socket
    import subprocess

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
        d = np.abs(gout[:, :wl[0]].astype(np.int32) - want[:, :wl[0]].astype(np.int32))
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


def measured_traffic(workload_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    passes (profiles/pmc_latest.json: FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950, + WRITE_SIZE), when they were taken on this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        if d.get("workload_key") == workload_key:
            return d["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist:
        dist.barrier()
    from rtlsdr_amd import synth
    from rtlsdr_amd.capi import ATAN_FAST, ATAN_LUT, ATAN_STD, RtlfmCfg
    from rtlsdr_amd.demod import GpuDemod

    atan = {"std": ATAN_STD, "fast": ATAN_FAST, "lut": ATAN_LUT}[a.atan]
    if a.boxcar:
        a.passes, a.fir9 = 0, 0
    D = a.boxcar if a.boxcar else 1 << a.passes
    fs = 2.4e6
    cfg = RtlfmCfg.default(downsample=D, downsample_passes=a.passes, comp_fir_size=9 if a.fir9 else 0,
                           custom_atan=atan, rate_out=int(fs / D), block_len=a.block_len,
                           max_blocks=a.blocks)
    S, NB, L = a.streams, a.blocks, a.block_len
    nsamp = NB * L // 2
    amp = 40.0 if a.atan == "fast" else 60.0  # -A fast overflows above |z| ~ 724
    iq = synth.fm_iq_u8_torch(S, nsamp, dev, fs=fs, dev_hz=75e3, amplitude=amp,
                              first_stream=rank * S)
    g = GpuDemod(cfg, S, local_rank)
    g.set_path(a.path)
    cap = g.result_cap(NB)
    out = torch.empty((S, cap), dtype=torch.int16, device=dev)
    out_len = torch.zeros(S, dtype=torch.int32, device=dev)

    def step():
        g.run_device(iq.data_ptr(), iq.stride(0), NB, out.data_ptr(), out.stride(0), out_len.data_ptr())

    def fence():
        g.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
            torch.cuda.synchronize()

    # a slice through the product before any timing (rank 0): the same handle type, same kernels;
    # the cpu_baseline leg below compares it with the CPU path's output for the same bytes
    gate = None
    if a.check and rank == 0 and not a.no_cpu_baseline and world == 1:
        cs, cb = min(S, 8), min(NB, 2)
        ccfg = RtlfmCfg.from_buffer_copy(bytes(cfg)); ccfg.max_blocks = cb
        sub = iq[:cs, :cb * L].contiguous()
        with GpuDemod(ccfg, cs, local_rank) as gc:
            gc.set_path(a.path)
            o, n = gc.run_torch(sub); gc.sync()
        gate = (ccfg, sub.cpu().numpy(), o.cpu().numpy(), n.cpu().numpy())

    # Untimed: after idle the GPU needs ~50 launches (~50 ms) to reach its steady clock.  If the
    # caller asks for fewer warm-up steps than that, the difference is run first and reported
    # as config.prewarm_steps, so that the K timed steps always measure the steady state.
    prewarm = max(0, 100 - a.warmup)
    for _ in range(prewarm + a.warmup):
        step()
    fence()
    g.timing_enable(True)
    g.timing_read()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    front_ms, launches = g.timing_read()
    g.timing_enable(False)
    path_used = g.last_path
    n_devices, scatter = 1, None
    if dist:
        cdev = dev if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # n_gpus = the devices that actually joined (ranks sharing a device count once)
        seen = torch.zeros(max(1, torch.cuda.device_count()), dtype=torch.int32, device=cdev)
        seen[local_rank] = 1
        dist.all_reduce(seen, op=dist.ReduceOp.MAX)
        n_devices = int(seen.sum().item())
        if a.scatter:
            scatter = time_scatter(dist, rank, world, dev, S, NB * L)

    if rank == 0:
        samples_per_step = world * S * nsamp
        value = samples_per_step * a.steps / elapsed / 1e6
        alg_bytes_per_sample = 2.0 + 2.0 / D  # u8 I + u8 Q in, int16 PCM out at 1/D
        launch_ms = front_ms / max(launches, 1)
        achieved = alg_bytes_per_sample * S * nsamp / (launch_ms * 1e-3) / 1e9 if launches else None
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
            "dtype": "int16/int32 fixed point (fp64 atan2)",
            "data": "synthetic",
            "config": {
                "workload": f"rtl_fm -M fm -s {int(fs / D)} " + ("" if a.boxcar else f"-F {9 if a.fir9 else 0} ")
                            + f"-A {a.atan}: {S} streams/GPU x {NB} buffers x {L} B u8 IQ @2.4 MS/s, "
                            + (f"low_pass boxcar /{D}" if a.boxcar else f"{a.passes}x fifth_order (/{D})")
                            + " + polar discriminant -> int16 PCM",
                "streams_per_gpu": S, "buffers_per_step": NB, "block_len": L, "passes": a.passes,
                "path": {1: "staged", 2: "fused"}.get(path_used, str(path_used)),
                "parallelism": f"streams sharded {S}/GPU over {n_devices} GPU(s), {world} rank(s), no data-path collective",
                "ranks": world,
                "requested_gpus": int(os.environ.get("RTLFM_BENCH_REQUESTED_GPUS", a.gpus)),
                "prewarm_steps": prewarm,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1) if achieved else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                "traffic": measured_traffic(f"S{S}_NB{NB}_L{L}_P{a.passes}_F{a.fir9}_{a.atan}"),
                "kernel": "decimating front end (convert+rotate+fifth_order[+fir9+discriminant])",
                "launch_ms": round(launch_ms, 4),
                "algorithmic_bytes_per_sample": alg_bytes_per_sample,
            },
        }
        if not a.no_cpu_baseline and world == 1:
            cs = min(S, os.cpu_count() or 1)
            sample = iq[:cs, :min(NB, 2) * L].contiguous().cpu().numpy()
            res["cpu_baseline"] = cpu_baseline(cfg, sample, a.cpu_seconds, gate)
            res["cpu_baseline"]["parity_checked"] = gate is not None
        else:
            res["cpu_baseline"] = None
        if scatter:
            res["scatter"] = scatter
        print(json.dumps(res))
    g.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
