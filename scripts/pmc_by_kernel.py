#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per (kernel, counter): mean per dispatch and dispatch count.
    python scripts/pmc_by_kernel.py <dir-with-*_counter_collection.csv> [name-filter]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if flt and flt not in name:
                continue
            import re
            m = re.search(r"(\w+)(<[^>]*>)?\(", name.replace("(anonymous namespace)::", ""))
            short = (m.group(1) + (m.group(2) or "")) if m else name[:60]
            key = (short[:60], row.get("Counter_Name", ""))
            acc[key][0] += float(row.get("Counter_Value", 0) or 0)
            acc[key][1] += 1
for (name, ctr), (tot, n) in sorted(acc.items()):
    print(f"{name:62s} {ctr:28s} dispatches {n:5d}  mean {tot / max(n, 1):16.1f}")
