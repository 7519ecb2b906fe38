#!/usr/bin/env python3
"""Static look at one kernel of the compiled library: instruction mix and register use.

    python tools/isa_count.py k_boxcar_scan [file.s]

Without a file the library's HIP source is compiled to gfx950 assembly in a temp dir."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly(src="rtlfm_hip.hip"):
    d = tempfile.mkdtemp()
    out = os.path.join(d, "k.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-w",
                           "--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "rtlsdr_amd", "csrc", src)])
    return out


def main():
    pat = sys.argv[1]
    path = sys.argv[2] if len(sys.argv) > 2 else assembly(os.environ.get("ISA_SRC", "rtlfm_hip.hip"))
    s = open(path).read()
    for m in re.finditer(r"^(_Z\S*" + re.escape(pat) + r"\S*):", s, re.M):
        name = m.group(1)
        body = s[m.end():s.index(".end_amdhsa_kernel", m.end())]
        code = body[:body.rindex("s_endpgm")]
        ins = []
        for line in code.splitlines():
            t = line.strip()
            if not t or t[0] in ".;" or t.endswith(":"):
                continue
            ins.append(t.split()[0])
        c = collections.Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        print(f"{name}\n  static instructions {len(ins)}: VALU {valu}, LDS {lds}, "
              f"SALU/other {len(ins) - valu - lds}")
        for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size"):
            mm = re.search(r"\.amdhsa_" + key + r"\s+(\S+)", body)
            if mm:
                print(f"  {key} {mm.group(1)}")
        mm = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", s)
        if mm:
            print("  num_vgpr", mm.group(1))
        print("  " + ", ".join(f"{k} {v}" for k, v in c.most_common(45)))


if __name__ == "__main__":
    main()
