#!/usr/bin/env python3
"""Prints 'name avg_us calls' for the kernels of a rocprofv3 --stats run (newest *kernel_stats.csv under DIR) whose
name contains FILTER.   usage: kstats.py DIR [FILTER]"""
import csv, glob, os, sys
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if not fs:
    sys.exit("no kernel_stats.csv under " + d)
out = []
for row in csv.DictReader(open(fs[-1])):
    n = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if flt in n:
        out.append(f"{n}={float(row['AverageNs'])/1e3:.1f}us")
print("  ".join(out))
