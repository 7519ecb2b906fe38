#!/usr/bin/env python3
"""Lint of the product's gfx950 ISA: nothing may touch a DOT result too early.

gfx90a and later do not interlock the result of v_dot2 / v_dot4 / v_dot8 against the VALU instructions right behind it.
LLVM's hazard recogniser (GCNHazardRecognizer::checkVALUHazards) keeps
  * a VALU that READS the result - any opcode but the same dot reading it as its accumulator - 3 wait states behind,
  * a VALU of another opcode that WRITES the same register 2 wait states behind,
for the dots it emits itself.  It does not look inside a line of inline assembly, and the kernels have such lines
(dot4_first, dot2_pair, chain_step): tools/dot_hazard_probe.hip shows v_dot4_i32_i8 wrong EVERY time when read at once.
This script walks the assembly of every kernel of both translation units and reports any read or write inside
the window, wherever the dot came from.

    python tools/check_dot_hazard.py [file.s ...]      exit 1 if any hazard (compiles the two units if no file is given)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ["rtlsdr_amd/csrc/rtlfm_hip.hip", "rtlsdr_amd/csrc/rtlpower_hip.hip", "rtlsdr_amd/csrc/rtlfm_place.hip"]
READ_WAIT, WRITE_WAIT = 3, 2

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(operand):
    out = set()
    for m in REG.finditer(operand):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_ops(text):
    text = text.split(";")[0].split("//")[0].strip()
    parts = text.split(None, 1)
    if not parts:
        return "", []
    ops = []
    if len(parts) > 1:
        depth, cur = 0, ""
        for ch in parts[1]:
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(cur.strip()); cur = ""
            else:
                cur += ch
        if cur.strip():
            ops.append(cur.strip())
    return parts[0], ops


def is_valu(op):
    return op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane")) or op.startswith(("v_readlane", "v_readfirstlane"))


def functions(asm):
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", asm, re.M):
        try:
            end = asm.index(".Lfunc_end", m.end())
        except ValueError:
            continue
        yield m.group(1), asm[m.end():end]


def check(body):
    ins = []
    for ln in body.split("\n"):
        t = ln.strip()
        if not t or t.startswith((";", "//", ".")) or t.endswith(":"):
            continue
        ins.append(t)
    problems = []
    ndots = 0
    for i, t in enumerate(ins):
        op, ops = split_ops(t)
        if not re.match(r"v_dot\d", op) or not ops:
            continue
        ndots += 1
        dst = regs(ops[0])
        waited = 0
        for j in range(i + 1, min(i + 8, len(ins))):
            op2, ops2 = split_ops(ins[j])
            if waited >= READ_WAIT:
                break
            if op2 == "s_nop":
                waited += int(ops2[0], 0) + 1
                continue
            if op2.startswith("v_") and ops2:
                same = op2 == op
                # destination(s): first operand (v_cmp writes vcc / sgprs; VOP3 with an sdst has two)
                wr = regs(ops2[0])
                rd_ops = ops2[1:]
                for k, o in enumerate(rd_ops):
                    r = regs(o)
                    if r & dst:
                        src_c = same and k == 2
                        if not src_c:
                            problems.append(f"'{t}' -> '{ins[j]}' reads the result after {waited} wait state(s) (needs {READ_WAIT})")
                if (wr & dst) and not same and waited < WRITE_WAIT:
                    problems.append(f"'{t}' -> '{ins[j]}' overwrites the result after {waited} wait state(s) (needs {WRITE_WAIT})")
                if wr & dst:
                    break  # redefined: later readers see the new value
            elif ops2 and any(regs(o) & dst for o in ops2) and op2.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")):
                pass  # memory instructions reading a VALU result are interlocked
            waited += 1
    return ndots, problems


def compile_units(defs):
    outs = []
    td = tempfile.mkdtemp(prefix="dotlint")
    procs = []
    for u in UNITS:
        out = os.path.join(td, os.path.basename(u) + ".s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", *defs, os.path.join(ROOT, u), "-o", out]
        procs.append((out, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    for out, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"hipcc failed for {out}")
        outs.append(out)
    return outs


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    files = [a for a in sys.argv[1:] if not a.startswith("-")]
    if not files:
        files = compile_units(defs)
    bad = nk = nd = 0
    for f in files:
        asm = open(f).read()
        for name, body in functions(asm):
            n, pr = check(body)
            nk += 1; nd += n
            if pr:
                bad += 1
                print(f"{name}: {len(pr)} hazard(s)")
                for p in pr[:4]:
                    print("    " + p)
    print(f"{nk} functions, {nd} dot instructions, {bad} function(s) with a dot hazard")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
