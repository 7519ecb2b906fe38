#!/usr/bin/env python3
"""Which quarter does an allocation of a given SIZE land in?  Twelve 4 GiB references are sorted into classes by probing
them pairwise (a pair runs slow exactly when both lie in one class, tools/placement_classes.py); then, for every size,
a sequence of blocks is allocated and each block classified against one representative per class."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtlsdr_amd.capi import load  # noqa: E402

lib = load()


def malloc(n):
    p = C.c_void_p()
    assert lib.rtlfm_gpu_malloc(0, n, C.byref(p)) == 0, n
    return p


def same(inp, inp_bytes, out, out_bytes):
    rd, rw = C.c_double(), C.c_double()
    r = lib.rtlfm_gpu_placement_probe(0, inp, inp_bytes, out, out_bytes, C.byref(rd), C.byref(rw))
    return r == 0


def main():
    G = 1 << 30
    refs = [malloc(4 * G) for _ in range(14)]
    classes = []  # list of lists of ref indices
    for i, p in enumerate(refs):
        for cl in classes:
            if same(refs[cl[0]], 4 * G, p, 4 * G):
                cl.append(i)
                break
        else:
            classes.append([i])
    print("classes of the 4 GiB references (allocation order):", classes)
    reps = [refs[cl[0]] for cl in classes]
    names = "ABCDEFGH"
    for mb in (64, 268, 512, 1024, 2048, 4096):
        n = mb << 20
        blocks, line = [], ""
        for k in range(20 if mb <= 1024 else 12):
            b = malloc(n)
            blocks.append(b)
            tag = "?"
            for ci, rp in enumerate(reps):
                if same(rp, 4 * G, b, n):
                    tag = names[ci]
                    break
            line += tag
        print(f"{mb:5d} MiB blocks, in allocation order: {line}", flush=True)
        for b in blocks:
            lib.rtlfm_gpu_free(b)
    # and once more with the blocks of a size kept while the next size is allocated
    for p in refs:
        lib.rtlfm_gpu_free(p)


if __name__ == "__main__":
    main()
