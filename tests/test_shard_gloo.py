"""N>1 path on CPU: world-size-2 (and 3) gloo jobs exercising the stream
sharding, the root->ranks scatter and the results gather used by bench.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from cases import CASES, make_cfg
from rtlsdr_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_range_is_a_partition():
    for n in (1, 2, 7, 256, 4096, 32768):
        for w in (1, 2, 3, 4, 8):
            got = []
            for r in range(w):
                a, b = shard.stream_range(n, w, r)
                assert 0 <= a <= b <= n
                got += list(range(a, b))
                assert (b - a) in (n // w, n // w + 1)
            assert got == list(range(n))
    assert shard.stream_range(32768, 8, 3) == (12288, 16384)  # BASELINE config 5: 4096 / GPU
    assert shard.owner_of(4096, 32768, 8) == 1


@pytest.mark.parametrize("world,nstreams", [(2, 5), (2, 8), (3, 7)])
def test_scatter_compute_gather_matches_single_process(tmp_path, oracle_lib, world, nstreams):
    out = tmp_path / "r.npz"
    port = 29500 + (os.getpid() + world * 7 + nstreams) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(out), str(nstreams)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    ov, _ = [(o, s) for n, o, s in CASES if n == "c2_p4_std"][0]
    cfg = make_cfg(ov, 8192, 3)
    want, want_len, _ = oracle_lib.run_batch(cfg, z["iq"], nthreads=2)
    assert np.array_equal(z["lens"], want_len)
    assert np.array_equal(z["out"], want)


def test_world_size_8_at_config_5_population(tmp_path, oracle_lib):
    """BASELINE configs[4]: 32768 streams sharded 4096 per GPU over eight ranks.  No eight-GPU node is ours to launch,
    so the part that needs no hardware runs here at the real population: eight gloo ranks, stream_range /
    scatter_streams / gather_results over 32768 streams (one 512-byte buffer each - the plumbing, not the DSP, is
    what scales with the population), every rank demodulating exactly its 4096, the gathered result equal to one
    process's."""
    nstreams, L, nb = 32768, 512, 1
    out = tmp_path / "r8.npz"
    port = 29500 + (os.getpid() * 5 + 811) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(out), str(nstreams), str(L), str(nb)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    assert int(z["world"]) == 8
    assert [tuple(x) for x in z["ranges"]] == [(4096 * k, 4096 * (k + 1)) for k in range(8)]
    ov, _ = [(o, s) for n, o, s in CASES if n == "c2_p4_std"][0]
    cfg = make_cfg(ov, L, nb)
    want, want_len, _ = oracle_lib.run_batch(cfg, z["iq"], nthreads=4)
    assert z["out"].shape[0] == nstreams
    assert np.array_equal(z["lens"], want_len)
    assert np.array_equal(z["out"], want)
