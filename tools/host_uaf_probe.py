#!/usr/bin/env python3
"""tools/host_uaf_probe.py — who writes into host memory it does not own?  (LAB.md I.21)

The one "parity mismatch" of rounds 5 and 6 is a 920-byte HOST array (460 int16 results of one stream of the p6firrdc /
11776-byte case) that changes after it has been compared and found right: the 16-bit word at byte 152 one less, the four
bytes at 888 zero - the footprint of a reference-count release and a cleared field of some freed native object whose block
the allocator handed to numpy.  This probe repeats the launch pattern of tests/test_soak_gpu.py (handle, upload, run_torch,
download, handle destroyed) with canaries - host arrays of exactly that size holding a pattern - and reports every canary
that changes, together with the host addresses of the runtime objects (streams, events, the handle) that died with the
handles just before.  Modes change ONE thing each, so that a same-box comparison names the owner:

    --mode base          the soak's pattern with the library as it is: a handle's streams go back to a pool (stream_pool.h)
    --mode destroy       RTLFM_DESTROY_STREAMS=1: round 5's behaviour, hipStreamDestroy when a handle goes - the canaries are hit
    --mode syncall       destroy + torch.cuda.synchronize() before the handle goes (every stream has consumed its waits): hit too
    --mode norelease     destroy, but no rtlfm_gpu_release_to / _wait_for: run_device + rtlfm_gpu_sync only - no hit
    --mode onehandle     one handle for all launches (reset between them): nothing is destroyed - no hit
(round 6, one box, profiles/r06_host_uaf_probe_box7.txt - there `base` still destroyed and `keepstreams` leaked the streams.)

    python tools/host_uaf_probe.py --mode base --launches 200000 [--canaries 64]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="base", choices=["base", "destroy", "syncall", "norelease", "onehandle"])
    ap.add_argument("--launches", type=int, default=100000)
    ap.add_argument("--canaries", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=240.0)
    a = ap.parse_args()
    if a.mode in ("destroy", "syncall", "norelease"):
        os.environ["RTLFM_DESTROY_STREAMS"] = "1"
    import faulthandler
    faulthandler.enable()
    import numpy as np
    import torch

    from cases import make_cfg
    from rtlsdr_amd import synth
    from rtlsdr_amd.capi import RtlfmCfg
    from rtlsdr_amd.demod import GpuDemod

    L, nb, ns = 512 * 23, 5, 5
    cfg = make_cfg(dict(downsample=64, downsample_passes=6, comp_fir_size=9, dc_block_raw=1), L, nb)
    iq = synth.fm_iq_u8(ns, L // 2 * nb, seed=7, fs=2.4e6, dev_hz=75e3, amplitude=50.0)
    first = None
    canaries, recent, hits, changed_results, kept = [], [], [], 0, []
    t0 = time.time()
    g_keep = GpuDemod(RtlfmCfg.from_buffer_copy(bytes(cfg)), ns, 0) if a.mode == "onehandle" else None
    n = 0
    for n in range(1, a.launches + 1):
        g = g_keep or GpuDemod(RtlfmCfg.from_buffer_copy(bytes(cfg)), ns, 0)
        if g_keep:
            g.reset()
        d = torch.from_numpy(iq).cuda()
        if a.mode == "norelease":
            cap = g.result_cap(nb)
            o = torch.empty((ns, cap), dtype=torch.int16, device="cuda")
            ln = torch.zeros(ns, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            g.run_device(d.data_ptr(), d.stride(0), nb, o.data_ptr(), o.stride(0), ln.data_ptr())
        else:
            o, ln = g.run_torch(d)
        g.sync()
        oh = o.cpu().numpy(); lh = ln.cpu().numpy()
        res = [oh[s, :lh[s]].copy() for s in range(ns)]
        objs = {k: g.get_option(k) for k in ("dbg_handle", "dbg_own_stream", "dbg_tail_stream", "dbg_event0", "dbg_event1", "dbg_event2",
                                             "dbg_event3", "dbg_event4", "dbg_event5")}
        if a.mode == "syncall":
            torch.cuda.synchronize()
        if not g_keep:
            g.close()
        recent.append(objs)
        del recent[:-8]
        if first is None:
            first = [r.copy() for r in res]
        # the results of the launch before this one once more, a whole launch after their handle has gone
        for held in kept:
            for s_, (x, y) in enumerate(zip(held, first)):
                if not np.array_equal(x, y):
                    at = np.flatnonzero(x != y)
                    changed_results += 1
                    print("RESULT ARRAY CHANGED", n, s_, hex(int(x.ctypes.data)), (at * 2).tolist(), x[at].tolist(), y[at].tolist(), flush=True)
                    x[:] = y
        kept.append(res)
        del kept[:-3]
        for c in canaries:
            if not (c == 0x5A5A).all():
                at = np.flatnonzero(c != 0x5A5A)
                addr = int(c.ctypes.data)
                near = [(k, hex(v), addr - v) for ob in recent for k, v in ob.items() if v and -256 <= addr - v <= 2048]
                hits.append((n, hex(addr), (at * 2).tolist(), [hex(int(x) & 0xffff) for x in c[at]], near))
                print("CANARY", hits[-1], flush=True)
                c[:] = 0x5A5A
        del canaries[:-a.canaries]
        canaries.extend(np.full(460, 0x5A5A, dtype=np.int16) for _ in range(a.canaries))
        if time.time() - t0 > a.seconds:
            break
    print(f"RESULT mode={a.mode}: {n} launches in {time.time() - t0:.0f} s, {len(hits)} canaries written by someone else, "
          f"{changed_results} result sets that differ from the first launch's", flush=True)


if __name__ == "__main__":
    main()
