#!/usr/bin/env python3
"""Per-pattern LDS counters from a rocprofv3 --pmc run of build_ablate/lds_probe (tools/lds_probe.hip)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"]][r["Counter_Name"]] = float(r["Counter_Value"])
n = 4096 * 2000
for k, v in d.items():
    print("%-28s per iteration: LDS instr %6.1f  active %7.1f  conflict cycles %7.1f  wave cycles %8.1f" % (
        k[:28], v.get("SQ_INSTS_LDS", 0) / n, v.get("SQ_ACTIVE_INST_LDS", 0) / n,
        v.get("SQ_LDS_BANK_CONFLICT", 0) / n, v.get("SQ_WAVE_CYCLES", 0) / n))
