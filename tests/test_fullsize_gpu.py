"""Parity at BASELINE.json's full sizes, through size-independent properties:
the oracle cannot process gigabytes in a test, so at full size the fused
kernel is compared with (a) the staged kernels on every sample, (b) the oracle
on a sample of streams end to end (all buffers, so carried state is exercised
through the whole run), (c) itself under run splitting and stream permutation."""
import numpy as np
import pytest

import golden_util as gu
from cases import make_cfg
from rtlsdr_amd import synth
from rtlsdr_amd.capi import RESAMPLE_ARBITRARY, RtlfmCfg

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _run(cfg, iq, path, splits=None, options=None):
    from rtlsdr_amd.demod import GpuDemod
    ns = iq.shape[0]
    L = int(cfg.block_len)
    nb = iq.shape[1] // L
    outs, lens = [], []
    with GpuDemod(cfg, ns, 0, options=options) as g:
        g.set_path(path)
        for (b0, b1) in (splits or [(0, nb)]):
            o, n = g.run_torch(iq[:, b0 * L:b1 * L] if (b0, b1) != (0, nb) else iq)
            g.sync()
            outs.append(o); lens.append(n)
        states = {s: g.state_get(s) for s in (0, ns // 2, ns - 1)}
        used = g.last_path
    return outs, lens, states, used


def _concat(outs, lens):
    n0 = [int(l[0]) for l in lens]
    return torch.cat([o[:, :n] for o, n in zip(outs, n0)], dim=1)


def test_c2_full_size_256_streams(oracle_lib):
    """configs[1]: 256 streams x 64 buffers x 262144 B, 4 passes + polar_discriminant (4 GiB of IQ)."""
    S, NB, L = 256, 64, 262144
    cfg = RtlfmCfg.default(downsample=16, downsample_passes=4, rate_out=150000, block_len=L, max_blocks=NB)
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, torch.device("cuda", 0))
    fo, fl, fst, used = _run(cfg, iq, 2)
    assert used == 2
    so, sl, sst, used1 = _run(cfg, iq, 1)
    assert used1 == 1
    n = int(fl[0][0])
    assert n == NB * (L // 2 // 16) and torch.equal(fl[0], sl[0])
    assert torch.equal(fo[0][:, :n], so[0][:, :n]), "fused != staged at full size"
    del so
    # oracle end to end on sampled streams (all 64 buffers => carried state through the run)
    for s in (0, 101, 255):
        want, st = oracle_lib.run_stream(cfg, iq[s].cpu().numpy())
        got = fo[0][s, :n].cpu().numpy()
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() <= 1e-4, (s, int(d.max()), int((d != 0).sum()))
        if s in fst:
            assert gu.state_dict(fst[s], False) == gu.state_dict(st, False)
    # splitting the run leaves every sample and the state unchanged
    po_, pl_, pst, _ = _run(cfg, iq, 2, splits=[(0, 16), (16, 17), (17, 64)])
    assert torch.equal(_concat(po_, pl_), fo[0][:, :n])
    for s in fst:
        assert gu.state_dict(pst[s], False) == gu.state_dict(fst[s], False)
    # streams are independent: permuting them permutes the output
    perm = torch.randperm(S, device=iq.device)
    qo, ql, _, _ = _run(cfg, iq[perm].contiguous(), 2)
    assert torch.equal(qo[0][:, :n], fo[0][perm, :n])


def test_c3_full_size_4096_nbfm_streams(oracle_lib):
    """configs[2] / one GPU's share of configs[4]: 4096 NBFM streams at 1.024 MS/s, /64 + FIR9 +
    deemph + arbitrary_resample 16k -> 22.05k, one 262144-B buffer each (1 GiB) per launch,
    eight launches with carried state."""
    S, NB, L = 4096, 8, 262144
    cfg = RtlfmCfg.default(downsample=64, downsample_passes=6, comp_fir_size=9, deemph=1, deemph_a=2,
                           rate_out=16000, rate_out2=22050, resampler=RESAMPLE_ARBITRARY,
                           block_len=L, max_blocks=1)
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, torch.device("cuda", 0), fs=1.024e6, dev_hz=2.5e3)
    launches = [(b, b + 1) for b in range(NB)]
    fo, fl, fst, used = _run(cfg, iq, 2, splits=launches)
    assert used == 2
    so, sl, sst, _ = _run(cfg, iq, 1, splits=launches)
    per = 2048 * 22050 // 16000
    assert int(fl[0][0]) == per == 2822
    f = _concat(fo, fl); s_ = _concat(so, sl)
    assert torch.equal(f, s_), "fused+tail != staged at full size"
    for s in (0, 1, 2047, 4095):
        want, st = oracle_lib.run_stream(cfg, iq[s].cpu().numpy())
        got = f[s].cpu().numpy()
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() <= 1e-4, (s, int(d.max()))
    for s in fst:
        assert gu.state_dict(fst[s], False) == gu.state_dict(sst[s], False)


def test_c5_stream_count_on_one_gpu(oracle_lib):
    """configs[4]'s whole population (32768 NBFM streams) on one GPU, one 262144-B buffer each
    (8 GiB of IQ): fused front end + tail against the staged kernels on every sample, and against
    the oracle on sampled streams."""
    S, NB, L = 32768, 1, 262144
    cfg = RtlfmCfg.default(downsample=64, downsample_passes=6, comp_fir_size=9, deemph=1, deemph_a=2,
                           rate_out=16000, rate_out2=22050, resampler=RESAMPLE_ARBITRARY,
                           block_len=L, max_blocks=1)
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, torch.device("cuda", 0), fs=1.024e6, dev_hz=2.5e3)
    fo, fl, fst, used = _run(cfg, iq, 2)
    assert used == 2
    so, sl, sst, _ = _run(cfg, iq, 1)
    n = int(fl[0][0])
    assert n == 2822 and torch.equal(fl[0], sl[0])
    assert torch.equal(fo[0][:, :n], so[0][:, :n])
    for s in (0, 4095, 4096, 20000, 32767):
        want, st = oracle_lib.run_stream(cfg, iq[s].cpu().numpy())
        got = fo[0][s, :n].cpu().numpy()
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() <= 1e-4, (s, int(d.max()))
    for s in fst:
        assert gu.state_dict(fst[s], False) == gu.state_dict(sst[s], False)


def test_default_buffer_size_many_buffers(oracle_lib):
    """rtl_fm's default 16384-B buffers (every 2nd tile starts a buffer): 64 streams x 512 buffers."""
    S, NB, L = 64, 512, 16384
    cfg = RtlfmCfg.default(downsample=16, downsample_passes=4, comp_fir_size=9, rate_out=150000,
                           block_len=L, max_blocks=NB)
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, torch.device("cuda", 0))
    fo, fl, fst, used = _run(cfg, iq, 2)
    so, sl, _, _ = _run(cfg, iq, 1)
    n = int(fl[0][0])
    assert used == 2 and torch.equal(fo[0][:, :n], so[0][:, :n])
    want, st = oracle_lib.run_stream(cfg, iq[S - 1].cpu().numpy())
    got = fo[0][S - 1, :n].cpu().numpy()
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-4
    assert gu.state_dict(fst[S - 1], False) == gu.state_dict(st, False)


def test_north_star_shape_4096_streams(oracle_lib):
    """north_star's target shape: 4096 batched 2.4 MS/s streams through the /16 FM path, 1 and then 3
    more 262144-B buffers each (one wave per stream and buffer run, no second wave per slot)."""
    S, NB, L = 4096, 4, 262144
    cfg = RtlfmCfg.default(downsample=16, downsample_passes=4, rate_out=150000, block_len=L, max_blocks=NB)
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, torch.device("cuda", 0))
    fo, fl, fst, used = _run(cfg, iq, 2, splits=[(0, 1), (1, 4)])
    assert used == 2
    got = _concat(fo, fl)
    n = got.shape[1]
    assert n == NB * (L // 2 // 16)
    so, sl, sst, used1 = _run(cfg, iq, 1)
    assert used1 == 1 and torch.equal(got, so[0][:, :n]), "fused (1 + 3 buffers) != staged (4 buffers)"
    del so
    for s in (0, 2048, 4095):
        want, st = oracle_lib.run_stream(cfg, iq[s].cpu().numpy())
        d = np.abs(got[s].cpu().numpy().astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() <= 1e-4, (s, int(d.max()))
        assert gu.state_dict(fst[s], False) == gu.state_dict(st, False)
    # ONE buffer per stream and launch - what a live capture hands over (src/rtl_fm.c:1339-1343): each
    # buffer is cut into segments inside the buffer (2 waves per stream by default, 4 when asked)
    for opts in (None, dict(fused_waves=16384)):
        po_, pl_, pst, used = _run(cfg, iq, 2, splits=[(0, 1), (1, 2), (2, 3), (3, 4)], options=opts)
        assert used == 2
        assert torch.equal(_concat(po_, pl_), got), f"one buffer per launch, {opts}"
        for s in fst:
            assert gu.state_dict(pst[s], False) == gu.state_dict(fst[s], False)


def test_wbfm_shape_full_size(oracle_lib):
    """rtl_fm -M wbfm at scale: 1024 streams x 16 x 262144 B, boxcar /6 (not dividing the buffer),
    -A fast, deemph, low_pass_real 170k -> 32k: the prefix-sum front end and the replay pass that
    feeds the resampler against the staged kernels + sequential filter on every sample, and
    against the oracle on sampled streams."""
    import os
    from rtlsdr_amd.capi import ATAN_FAST, RESAMPLE_LOW_PASS_REAL
    S, NB, L = 1024, 16, 262144
    cfg = RtlfmCfg.default(downsample=6, custom_atan=ATAN_FAST, rate_out=170000, deemph=1, deemph_a=13, rate_out2=32000,
                           resampler=RESAMPLE_LOW_PASS_REAL, block_len=L, max_blocks=NB)
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, torch.device("cuda", 0), fs=1.02e6, dev_hz=75e3, amplitude=30.0)
    fo, fl, fst, used = _run(cfg, iq, 2)
    assert used == 2
    so, sl, sst, used1 = _run(cfg, iq, 1, options=dict(deemph_sequential=1))
    assert used1 == 1 and torch.equal(fl[0], sl[0])
    n = int(fl[0].max())
    mask = torch.arange(n, device=iq.device)[None, :] < fl[0][:, None]
    assert torch.equal(fo[0][:, :n][mask], so[0][:, :n][mask]), "fused boxcar + fused tail != staged + sequential"
    del so
    for s in (0, 512, 1023):
        want, st = oracle_lib.run_stream(cfg, iq[s].cpu().numpy())
        got = fo[0][s, :int(fl[0][s])].cpu().numpy()
        assert np.array_equal(got, want), s
        assert gu.state_dict(fst[s], False) == gu.state_dict(st, False)
    po_, pl_, pst, _ = _run(cfg, iq, 2, splits=[(0, 5), (5, 6), (6, 16)])
    tot = sum(int(l[0]) for l in pl_)
    assert tot == int(fl[0][0])
    assert torch.equal(torch.cat([o[0, :int(l[0])] for o, l in zip(po_, pl_)]), fo[0][0, :tot])
    for s in fst:
        assert gu.state_dict(pst[s], False) == gu.state_dict(fst[s], False)


def test_scanner_line_full_size(oracle_lib):
    """The everyday scanner line at configs[1]'s size: rtl_fm -M fm -s 12k -l 50 (boxcar /84 + power squelch) on 256
    streams x 64 buffers x 262144 B of keyed carriers - the boxcar front end in emit mode + the squelch / demod kernels
    (`last_path == 2`) against the staged kernels on every sample and the oracle on sampled streams, split runs."""
    S, NB, L = 256, 64, 262144
    cfg = RtlfmCfg.default(downsample=84, rate_out=12000, squelch_level=50, block_len=L, max_blocks=NB)
    dev = torch.device("cuda", 0)
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, dev, fs=1.008e6, dev_hz=2.5e3, amplitude=0.8)
    # key the carriers: every stream silent for its own stretch of whole and half buffers
    for s in range(0, S, 3):
        a = ((s * 7919) % (NB - 6)) * L + (L // 2 if s % 2 else 0)
        iq[s, a:a + 5 * L // 2] = 127
    fo, fl, fst, used = _run(cfg, iq, 2)
    assert used == 2
    so, sl, sst, used1 = _run(cfg, iq, 1)
    assert used1 == 1 and torch.equal(fl[0], sl[0])
    n = int(fl[0].max())
    mask = torch.arange(n, device=dev)[None, :] < fl[0][:, None]
    assert torch.equal(fo[0][:, :n][mask], so[0][:, :n][mask]), "emit-mode front end != staged at full size"
    del so
    for s in (0, 3, 129, 255):
        want, st = oracle_lib.run_stream(cfg, iq[s].cpu().numpy())
        got = fo[0][s, :int(fl[0][s])].cpu().numpy()
        assert got.shape == want.shape
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() <= 1e-4, (s, int(d.max()))
        if s in fst:
            assert gu.state_dict(fst[s], False) == gu.state_dict(st, False)
    po_, pl_, pst, _ = _run(cfg, iq, 2, splits=[(0, 7), (7, 8), (8, 64)])
    tot = sum(int(l[0]) for l in pl_)
    assert tot == int(fl[0][0])
    assert torch.equal(torch.cat([o[0, :int(l[0])] for o, l in zip(po_, pl_)]), fo[0][0, :tot])
    for s in fst:
        assert gu.state_dict(pst[s], False) == gu.state_dict(fst[s], False)


@pytest.mark.parametrize("front", ["p4", "box10"])
def test_odd_buffer_length_full_size(oracle_lib, front):
    """-W 40 (20480-byte buffers: two whole tiles and half a tile each) at scale: 1024 streams x 200 buffers = 4 GB of IQ
    through the partial-tile fifth_order kernels / the boxcar's continuous run, against the staged kernels on every
    sample, the oracle on sampled streams, and itself under run splitting."""
    S, NB, L = 1024, 200, 20480
    ov = dict(downsample=16, downsample_passes=4) if front == "p4" else dict(downsample=10, downsample_passes=0)
    cfg = RtlfmCfg.default(rate_out=150000, block_len=L, max_blocks=NB, **ov)
    iq = synth.fm_iq_u8_torch(S, NB * L // 2, torch.device("cuda", 0))
    fo, fl, fst, used = _run(cfg, iq, 2)
    assert used == 2
    so, sl, sst, used1 = _run(cfg, iq, 1)
    assert used1 == 1 and torch.equal(fl[0], sl[0])
    n = int(fl[0].max())
    mask = torch.arange(n, device=iq.device)[None, :] < fl[0][:, None]
    assert torch.equal(fo[0][:, :n][mask], so[0][:, :n][mask]), "one-launch front end != staged at full size"
    del so
    for s in (0, 512, 1023):
        want, st = oracle_lib.run_stream(cfg, iq[s].cpu().numpy())
        got = fo[0][s, :int(fl[0][s])].cpu().numpy()
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert got.shape == want.shape and d.max() <= 1 and (d != 0).mean() <= 1e-4, (s, int(d.max()))
        assert gu.state_dict(fst[s], False) == gu.state_dict(st, False)
    po_, pl_, pst, _ = _run(cfg, iq, 2, splits=[(0, 33), (33, 34), (34, 200)])
    tot = sum(int(l[0]) for l in pl_)
    assert tot == int(fl[0][0])
    assert torch.equal(torch.cat([o[0, :int(l[0])] for o, l in zip(po_, pl_)]), fo[0][0, :tot])


def test_rtl_power_fine_bins_full_size(oracle_lib):
    """frequency_range()'s fine-bin plans at scale: 2^17 bins, 64 tuning states x 16 reads x 262144 B (256 MiB of IQ per
    launch) through the transform over HBM - every accumulator equal to the oracle's on sampled states, reads split
    over two launches equal to one launch, and Parseval-like sanity: a state fed zeros-around-127 accumulates nothing
    but the window's DC leakage."""
    from rtlsdr_amd.capi import RtlpowerCfg
    from rtlsdr_amd.power import GpuPower
    S, NR, bin_e = 64, 16, 17
    L = 2 << bin_e
    cfg = RtlpowerCfg.default(bin_e=bin_e, window=1, buf_len=L)
    dev = torch.device("cuda", 0)
    iq = synth.fm_iq_u8_torch(S, NR * L // 2, dev, fs=2.048e6, dev_hz=50e3)
    iq[S - 1] = 127
    with GpuPower(cfg, S, 0) as g:
        g.scan_torch(iq); g.sync()
        one = [g.fetch(s) for s in (0, 31, S - 1)]
        full = torch.from_numpy(np.stack([g.fetch(s)[0] for s in range(S)]))
    with GpuPower(cfg, S, 0) as g:
        g.scan_torch(iq[:, :5 * L].contiguous()); g.scan_torch(iq[:, 5 * L:].contiguous()); g.sync()
        split = torch.from_numpy(np.stack([g.fetch(s)[0] for s in range(S)]))
    assert torch.equal(full, split)
    for (avg, n), s in zip(one, (0, 31, S - 1)):
        want, wn = oracle_lib.power_scan_batch(cfg, iq[s:s + 1].cpu().numpy(), nthreads=1)
        assert n == wn[0] == NR and np.array_equal(avg, want[0]), s
    assert int(one[2][0].sum()) == 0  # 127 is sample 0: nothing but zeros goes into the transform
