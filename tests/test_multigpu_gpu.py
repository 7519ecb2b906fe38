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
    # whole-job value: both ranks' samples over the max-over-ranks time
    per_step = 2 * 32 * 4 * 262144 // 2
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02
