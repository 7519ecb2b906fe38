#!/usr/bin/env python3
"""Where are the boundaries?  One 4 GiB input, then 268 MiB candidates with 4 GiB fillers between them, up to --gb of
device memory: for every candidate the bandwidth probe's read-only and read + write times against the input
(rtlfm_gpu_placement_probe).  A ratio near 1.12 = another quarter of the HBM, near 1.31 = the input's own."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd.capi import load  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=int, default=230)
    ap.add_argument("--pre", type=int, default=0, help="GiB allocated (and kept) before the input")
    a = ap.parse_args()
    lib = load()
    dev = torch.device("cuda:0")
    pre = torch.empty(a.pre << 30, dtype=torch.uint8, device=dev) if a.pre else None
    iq = torch.randint(0, 256, (4 << 30,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    print(f"input @{iq.data_ptr():#x}, free {free >> 30} GiB of {total >> 30}")
    keep, walked = [], 0
    cb = 268 << 20
    while walked < (a.gb << 30):
        p = C.c_void_p()
        if lib.rtlfm_gpu_malloc(0, cb, C.byref(p)) != 0:
            print("out of memory"); break
        rd, rw = C.c_double(), C.c_double()
        r = lib.rtlfm_gpu_placement_probe(0, iq.data_ptr(), iq.numel(), p, cb, C.byref(rd), C.byref(rw))
        print(f"walked {walked / 1e9:6.1f} GB  cand @{p.value:#x}  read {rd.value:.4f} ms  r+w {rw.value:.4f} ms  ratio {rw.value / rd.value:.3f}  apart={r}", flush=True)
        keep.append(p)
        f = C.c_void_p()
        if lib.rtlfm_gpu_malloc(0, 4 << 30, C.byref(f)) != 0:
            print("out of memory (filler)"); break
        keep.append(f)
        walked += cb + (4 << 30)
    for p in keep:
        lib.rtlfm_gpu_free(p)


if __name__ == "__main__":
    main()
