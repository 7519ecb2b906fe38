#!/usr/bin/env python3
"""Lint of the fused kernels' ISA: the next tile's loads must stay in flight.

Each k_fused instantiation issues the next tile's eight global_load_dwordx4 in
the middle of a tile and consumes them at the top of the next iteration.  The
register allocator sometimes splits the live range of a loaded register and
inserts `s_waitcnt vmcnt(N)` + v_mov right after the loads (or at the loop
bottom), which stalls the wave for a full HBM latency every tile.  This script
compiles rtlfm_hip.hip to assembly and reports, per kernel, every vmcnt wait
between the tile loads and the loop back-edge that forces one of them to
complete (vmcnt counts in order: a wait for N forces every operation that has
at least N younger ones).

    python tools/check_prefetch.py [-DNAME=VALUE ...] [rtlfm_hip.s]      exit 1 if any kernel stalls
(an assembly file that exists already - tests/test_isa_lint.py compiles once for both of its lints - is read instead of compiling)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(asm):
    for m in re.finditer(r"^(_ZN5rtlfm5fused7k_fusedI\w+):[^\n]*\n", asm, re.M):
        end = asm.index(".Lfunc_end", m.end())
        yield m.group(1), asm[m.end():end]


def instrs(body):
    out = []
    for ln in body.split("\n"):
        t = ln.strip()
        if not t or t.startswith((";", "//", ".p2align", ".loc", ".cfi")):
            continue
        out.append(t)
    return out


def check(name, body):
    ins = instrs(body)
    labels = {m.group(1): i for i, t in enumerate(ins) for m in [re.match(r"^(\.LBB\d+_\d+):", t)] if m}
    # the tile loop: the largest backward branch span
    best = None
    for i, t in enumerate(ins):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (i - labels[m.group(1)], labels[m.group(1)], i)
            if best is None or span > best:
                best = span
    if best is None:
        return ["no loop found"]
    _, lo, hi = best
    loads = [i for i in range(lo, hi) if ins[i].startswith("global_load_dwordx4")]
    if len(loads) < 8:
        return [f"expected 8 tile loads in the loop, found {len(loads)}"]
    loads = loads[-8:]
    problems = []
    younger = 0  # vmem operations issued after the first tile load
    for i in range(loads[0] + 1, hi + 1):
        t = ins[i]
        if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", t):
            younger += 1
        m = re.search(r"vmcnt\((\d+)\)", t)
        # a wait in the last instructions before the back-edge is the consuming wait of the
        # next iteration (rotated loop), not a stall
        if m and younger >= int(m.group(1)) and i > loads[0] and hi - i > 24:
            done = younger - int(m.group(1)) + 1
            problems.append(f"instr {i - lo} of {hi - lo}: '{t}' forces {min(done, 8)} tile load(s) to land "
                            f"{'right after issue' if i - loads[-1] < 20 else 'before the back-edge'}")
    return problems


def compile_asm(defs):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fm.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", *defs,
               os.path.join(ROOT, "rtlsdr_amd/csrc/rtlfm_hip.hip"), "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    given = [a for a in sys.argv[1:] if not a.startswith("-")]
    asm = open(given[0]).read() if given else compile_asm(defs)
    bad = 0
    for name, body in kernels(asm):
        m = re.match(r"_ZN5rtlfm5fused7k_fusedILi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
        tag = "P=%s fir9=%s std=%s mfma=%s rdc=%s pt=%s" % m.groups() if m else name
        pr = check(name, body)
        if pr and m and m.group(6) == "1":
            # the partial-tile kernels (-W n) are a by-road: reported, not counted
            print(f"{tag}: stall (partial-tile kernel, not counted)")
            for p in pr[:2]:
                print("    " + p)
        elif pr:
            bad += 1
            print(f"{tag}: STALL")
            for p in pr[:4]:
                print("    " + p)
        else:
            print(f"{tag}: ok")
    print(f"{bad} kernel(s) with a forced wait on the prefetch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
