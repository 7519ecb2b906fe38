"""N > 1 on hardware: world-size-2 jobs whose per-rank compute is the HIP
library.  Two devices + RCCL when the box has them; on a one-GPU box the same job runs with
both ranks on device 0 and gloo for the exchange, so the launcher, the sharding and the
library-under-two-processes path are exercised either way."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from cases import CASES, make_cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ndev():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("case,nstreams", [("c2_p4_std", 6), ("c1_boxcar10_fast", 5)])
def test_two_ranks_hip_library_matches_oracle(tmp_path, oracle_lib, case, nstreams):
    out = tmp_path / "r.npz"
    port = 29500 + (os.getpid() * 3 + nstreams) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker_gpu.py"), str(out), str(nstreams), case]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(out)
    assert int(z["multi"]) == int(_ndev() >= 2)
    ov, _ = [(o, s) for n, o, s in CASES if n == case][0]
    cfg = make_cfg(ov, 16384, 3)
    want, want_len, _ = oracle_lib.run_batch(cfg, z["iq"], nthreads=2)
    assert np.array_equal(z["lens"], want_len)
    for s in range(nstreams):
        assert np.array_equal(z["out"][s, :want_len[s]], want[s, :want_len[s]]), s


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` without torchrun must start the ranks itself and report
    the devices that joined."""
    two = _ndev() >= 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not two:
        env["RTLFM_BENCH_BACKEND"] = "gloo"  # both ranks on the one device
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--streams", "32", "--blocks", "4", "--no-cpu-baseline", "--scatter"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["config"]["ranks"] == 2
    assert d["n_gpus"] == (2 if two else 1)
    assert d["value"] > 0 and d["scaling"] == "weak"
    assert d["scatter"]["bytes_from_root"] > 0
    # what the 1/2/4/8 curve is read with: every rank's own step and launch time, whether its output got its placement,
    # and its PCIe-inclusive rate beside its pinned H2D rate (at eight GPUs the host side is the curve)
    pr = d["per_rank"]
    assert pr["world_size"] == 2 and len(pr["ms_per_step"]) == 2 and len(pr["output_apart"]) == 2
    assert len(pr["e2e_GB/s_in"]) == 2 and min(pr["e2e_GB/s_in"]) > 0 and min(pr["pinned_h2d_GB/s"]) > 0
    # whole-job value: both ranks' samples over the max-over-ranks time
    per_step = 2 * 32 * 4 * 262144 // 2
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02


def test_bench_gpus_8_launcher_at_world_size_eight():
    """`python bench.py --gpus 8` as the driver's scaling run starts it, on whatever this box has: with fewer than eight
    devices the eight ranks share device 0 and gloo carries the barrier, the max-over-ranks time and the all_gather of the
    per-rank report (RTLFM_BENCH_BACKEND=gloo exists for exactly this) - the launcher, the stream sharding 'n / GPU', the
    reduction and the JSON at world size 8, which no one-GPU box otherwise sees."""
    eight = _ndev() >= 8
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not eight:
        env["RTLFM_BENCH_BACKEND"] = "gloo"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
           "--streams", "16", "--blocks", "2", "--no-cpu-baseline", "--e2e", "0", "--sustain", "0", "--also", "0", "--ceiling", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["ranks"] == 8 and d["scaling"] == "weak"
    assert d["n_gpus"] == (8 if eight else 1)
    pr = d["per_rank"]
    assert pr["world_size"] == 8 and len(pr["ms_per_step"]) == 8 and len(pr["launch_ms"]) == 8 and len(pr["output_apart"]) == 8
    assert min(pr["ms_per_step"]) > 0 and max(pr["ms_per_step"]) <= d["ms_per_step"] + 0.001  # (the line rounds to 1 us, the ranks to 0.1 us)
    per_step = 8 * 16 * 2 * 262144 // 2  # weak scaling: every rank its own 16 streams
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02


@pytest.mark.parametrize("args", [
    ["--workload", "c2", "--streams", "16", "--blocks", "4"],
    ["--workload", "ns4096", "--streams", "64", "--blocks", "1"],
    ["--workload", "c1", "--streams", "16", "--blocks", "4"],
    ["--workload", "wbfm", "--streams", "16", "--blocks", "4"],
    ["--workload", "c3", "--streams", "64", "--blocks", "2"],
    ["--workload", "c4", "--streams", "8", "--blocks", "4"],
])
def test_bench_workloads_emit_the_contract_line(args):
    """Every BASELINE-shaped workload of bench.py at a small size: one JSON line with the contract's
    fields, a roofline object incl. the sustained leg, a cpu_baseline with its parity gate passed."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--pmc", "0", "--sustain", "0.2",
           "--cpu-seconds", "0.5"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and d["vs_baseline"] is None
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["peak"] == 8000.0 and 0 < ro["frac"] < 1 and ro["sustained"]["steps"] >= 4
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert d["cpu_baseline"]["parity_checked"] is True and d["cpu_baseline"]["kind"] in ("reference", "port")
    assert args[1] in d["config"]["workload"]
    if args[1] != "c4":
        assert d["e2e"]["value"] > 0
