#!/usr/bin/env python3
"""Which buffers run apart from which?  N inputs of 4 GiB allocated one after the other, M outputs of 268 MiB allocated
between them; the bandwidth probe's read + write / read-only ratio for every (input, output) pair
(rtlfm_gpu_placement_probe).  If memory falls into classes and a pair is slow exactly when both are in one class
(DESIGN.md section 3.1), the matrix is a block pattern: rows and columns sort into the classes."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd.capi import load  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inputs", type=int, default=40)
    ap.add_argument("--every", type=int, default=4, help="an output after every this many inputs")
    a = ap.parse_args()
    lib = load()
    ins, outs = [], []
    ib, ob = 4 << 30, 268 << 20
    for i in range(a.inputs):
        if i % a.every == 0:
            p = C.c_void_p()
            assert lib.rtlfm_gpu_malloc(0, ob, C.byref(p)) == 0
            outs.append((i, p))
        p = C.c_void_p()
        if lib.rtlfm_gpu_malloc(0, ib, C.byref(p)) != 0:
            break
        ins.append(p)
    print(f"{len(ins)} inputs of 4 GiB, {len(outs)} outputs (allocated before input {[i for i, _ in outs]})")
    print("rows: inputs in allocation order; columns: outputs; '.' = apart (ratio < 1.21), '#' = same class")
    for i, p in enumerate(ins):
        row, ratios = "", []
        for _, o in outs:
            rd, rw = C.c_double(), C.c_double()
            r = lib.rtlfm_gpu_placement_probe(0, p, ib, o, ob, C.byref(rd), C.byref(rw))
            row += "." if r == 1 else "#"
            ratios.append(rw.value / rd.value)
        print(f"in {i:2d} @{p.value:#x}  {row}   " + " ".join(f"{x:.2f}" for x in ratios), flush=True)
    for p in ins:
        lib.rtlfm_gpu_free(p)
    for _, o in outs:
        lib.rtlfm_gpu_free(o)


if __name__ == "__main__":
    main()
