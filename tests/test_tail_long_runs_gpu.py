"""Long runs through the audio tail (deemph_filter, then low_pass_real or arbitrary_resample): the
time-parallel and one-pass forms of the filter only engage from 2048 demodulated samples per
stream on, which the short buffers of the general sweep rarely reach.  Seeded random
configurations against the oracle: rates, filter constants, buffer counts, silent and
half-silent streams (the fall-back list), injected filter / resampler state, split launches.

RTLFM_TAIL_FUZZ=<n> widens the sweep to n seeds (default 10)."""
import os

import numpy as np
import pytest

import golden_util as gu
from cases import make_cfg
from rtlsdr_amd import capi, synth
from test_parity_gpu import assert_parity

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

NSEEDS = int(os.environ.get("RTLFM_TAIL_FUZZ", "10"))


def _cfg(rng):
    ov = dict(custom_atan=int(rng.choice([0, 1, 2])))
    kind = rng.random()
    if kind < 0.6:
        p_ = int(rng.integers(2, 7))
        ov.update(downsample=1 << p_, downsample_passes=p_, comp_fir_size=int(rng.choice([0, 9])))
        L = 262144 if p_ >= 5 else int(rng.choice([65536, 131072, 262144]))
    else:
        ov.update(downsample=int(rng.choice([2, 4, 6, 8, 10, 16, 32])), downsample_passes=0)   # divides every buffer
        L = int(rng.choice([61440, 122880, 245760]))  # multiples of 2 * lcm(2..16, 32) * ...: uniform counts for these D
    ov["rate_out"] = int(rng.choice([16000, 24000, 48000, 150000, 170000, 240000]))
    ov.update(deemph=1, deemph_a=int(rng.choice([2, 2, 3, 5, 8, 12, 13, 19, 30])))
    r = rng.random()
    if r < 0.4:
        ov.update(rate_out2=max(1, int(ov["rate_out"] * rng.choice([0.1882, 0.23, 0.4, 0.5, 0.75, 0.99999]))),
                  resampler=capi.RESAMPLE_LOW_PASS_REAL)
    elif r < 0.8:
        ov.update(rate_out2=int(ov["rate_out"] * rng.choice([1.0006, 1.1, 1.378125, 1.5, 1.9, 0.6])), resampler=capi.RESAMPLE_ARBITRARY)
    if rng.random() < 0.15:
        ov["dc_block_audio"] = 1
    if rng.random() < 0.1:
        ov["post_downsample"] = 2
    return ov, L


@pytest.mark.parametrize("seed", range(NSEEDS))
def test_long_run_tail_random(oracle_lib, seed):
    from rtlsdr_amd.demod import GpuDemod
    rng = np.random.default_rng(31000 + seed)
    ov, L = _cfg(rng)
    nb = int(rng.integers(2, 7))
    ns = int(rng.choice([3, 5, 9]))
    cfg = make_cfg(ov, L, nb)
    try:
        GpuDemod(cfg, ns, 0).close()
    except capi.RtlfmError as e:
        pytest.skip(f"rejected: {e} {ov}")
    amp = 20.0 if ov["custom_atan"] == 1 else 50.0
    if ov["custom_atan"] == 1 and ov["downsample_passes"] == 0:
        amp = max(2.0, min(20.0, 500.0 / ov["downsample"]))
    iq = synth.fm_iq_u8(ns, L // 2 * nb, seed=8100 + seed, fs=1.024e6, dev_hz=float(rng.choice([3e3, 20e3, 75e3])), amplitude=amp)
    if rng.random() < 0.7:
        iq[1] = 127
    if rng.random() < 0.7:
        b0 = int(rng.integers(0, nb))
        iq[2, b0 * L:(b0 + 1) * L] = 127
    st0 = oracle_lib.new_states(ns)
    for s in range(ns):
        st0[s].deemph_avg = int(rng.integers(-32768, 32768))
        if ov.get("resampler") == capi.RESAMPLE_LOW_PASS_REAL and ov.get("rate_out2"):
            st0[s].now_lpr = int(rng.integers(-100000, 100000))
            st0[s].prev_lpr_index = int(rng.integers(0, ov["rate_out"]))
    if rng.random() < 0.4:
        st0[ns - 1].deemph_avg = int(rng.choice([-70000, 40000, 1 << 20]))
    st_copy = [capi.RtlfmStreamState.from_buffer_copy(bytes(st0[s])) for s in range(ns)]
    want, want_len, wst = oracle_lib.run_batch(cfg, iq, states=st0, nthreads=4)
    cut = int(rng.integers(1, nb))
    for splits in (None, [(0, cut), (cut, nb)]):
        outs = [[] for _ in range(ns)]
        with GpuDemod(cfg, ns, 0) as g:
            for s in range(ns):
                g.state_set(s, st_copy[s])
            d = torch.from_numpy(iq).cuda()
            for b0, b1 in (splits or [(0, nb)]):
                o, n = g.run_torch(d[:, b0 * L:b1 * L].contiguous()); g.sync()
                o = o.cpu().numpy(); n = n.cpu().numpy()
                for s in range(ns):
                    outs[s].append(o[s, :n[s]].copy())
            sts = [g.state_get(s) for s in range(ns)]
        for s in range(ns):
            got = np.concatenate(outs[s])
            assert got.shape[0] == want_len[s], (ov, L, nb, splits, s, got.shape[0], int(want_len[s]))
            assert_parity(got, want[s, :want_len[s]], cfg, f"{ov} L={L} nb={nb} {splits} stream {s}")
            assert gu.state_dict(sts[s], False) == gu.state_dict(wst[s], False), (ov, splits, s)
