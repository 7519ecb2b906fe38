"""fused::plan_segments by itself (the planner had no direct unit test): how a front-end launch is cut into
waves, through the host-only export rtlfm_plan_segments - no GPU.  What every plan must satisfy, whatever rule of thumb
produced it: the segments of a stream partition its tiles in order, none is empty, and - because every segment but a
stream's first re-runs one warm-up tile - none is shorter than `min_tiles` unless the launch could not fill the GPU's
4096 wave slots otherwise."""
import ctypes as C

import pytest

from rtlsdr_amd import capi

WAVE_SLOTS = 256 * 4 * 4


def plan(nstreams, total_tiles, fifth=1, tail=0, target=0, min_tiles=0, tps=0, gss=0):
    lib = capi.load()
    segs = C.c_int()
    starts = (C.c_int * 4096)()
    capi.check(lib.rtlfm_plan_segments(nstreams, total_tiles, fifth, tail, target, min_tiles, tps, gss, C.byref(segs), starts, 4096),
               "rtlfm_plan_segments")
    return segs.value, list(starts[:segs.value + 1])


@pytest.mark.parametrize("tail", [0, 1])
@pytest.mark.parametrize("gss", [0, 15, 20])
@pytest.mark.parametrize("nstreams,total_tiles", [(1, 1), (1, 2), (1, 32), (5, 10), (40, 8), (256, 2048), (1024, 512), (4096, 32), (4096, 128),
                                                  (4096, 512), (32768, 32), (3, 7), (4095, 33), (8192, 1)])
def test_every_plan_is_an_ordered_partition(nstreams, total_tiles, gss, tail):
    segs, starts = plan(nstreams, total_tiles, tail=tail, gss=gss)
    assert segs >= 1 and starts[0] == 0 and starts[-1] == total_tiles
    lens = [b - a for a, b in zip(starts, starts[1:])]
    assert all(n >= 1 for n in lens), (starts,)
    # at least as many waves as fill the slots, where the tiles allow it at all
    if nstreams * total_tiles >= WAVE_SLOTS:
        assert nstreams * segs >= min(WAVE_SLOTS, nstreams), (segs,)
    # segments are not shorter than eight tiles (their warm-up tile: at most 1/8 on top) unless the launch is underfilled;
    # a uniform plan's LAST segment takes what is left (guided plans let it absorb the crumbs instead)
    if nstreams * (total_tiles // 8) >= WAVE_SLOTS and segs > 1:
        assert min(lens[:-1]) >= 8, (lens,)
    # guided plans hand out the long segments first (the last one may have absorbed up to min_tiles - 1 tiles of crumbs)
    if gss and segs > 1 and len(set(lens)) > 1:
        assert all(a >= b for a, b in zip(lens[:-1], lens[1:-1])) and lens[0] >= lens[-1] - 7, (lens,)


def test_the_rules_of_thumb_the_measurements_left():
    # north_star's live shape - 4096 streams x one 262144-byte buffer = 32 tiles: one wave per stream (LAB.md: 0.2050 against 0.2086 ms)
    assert plan(4096, 32)[0] == 1
    # four buffers per launch: two waves per stream
    assert plan(4096, 128)[0] == 2
    # the reference's own shape - one stream, one 16384-byte buffer = 2 tiles: a wave per tile (nothing else would use the GPU)
    assert plan(1, 2) == (2, [0, 1, 2])
    # in front of an audio tail the segments are shorter (24576 waves behind fifth_order passes, 20480 behind the boxcar)
    assert plan(4096, 128, fifth=1, tail=1)[0] == 6 and plan(4096, 128, fifth=0, tail=1)[0] == 5
    # an explicit segment length rules
    assert plan(7, 10, tps=3) == (4, [0, 3, 6, 9, 10])
    # the caller's wave count: one wave in all = one segment per stream
    assert plan(5, 10, target=1)[0] == 1


def test_bad_arguments():
    lib = capi.load()
    segs = C.c_int()
    starts = (C.c_int * 4)()
    assert lib.rtlfm_plan_segments(0, 10, 1, 0, 0, 0, 0, 0, C.byref(segs), starts, 4) < 0
    assert lib.rtlfm_plan_segments(5, 10, 1, 0, 0, 0, 1, 0, C.byref(segs), starts, 4) < 0  # ten segments do not fit four entries
    assert segs.value == 10
