#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per-dispatch averages for the k_fused kernel
(largest dispatches only, so the small parity-gate launch is left out)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("PMC_KERNEL", "k_fused")
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if pat not in r["Kernel_Name"]:
                continue
            rows[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for name in sorted(rows):
    per = [sum(v) for v in rows[name].values()]
    big = [x for x in per if x >= 0.5 * max(per)] if max(per) > 0 else per
    print(f"{name:28s} dispatches={len(big):3d} avg={sum(big)/len(big):.6g}")
