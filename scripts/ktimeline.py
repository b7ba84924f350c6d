#!/usr/bin/env python3
"""Prints the kernel timeline (start/end in us, queue) of one training step from a rocprofv3 --kernel-trace run.
   usage: ktimeline.py DIR [anchor-kernel-substring]"""
import csv, glob, os, sys
d = sys.argv[1]; anchor = sys.argv[2] if len(sys.argv) > 2 else "head_kernel"
fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if not fs:
    sys.exit("no kernel_trace.csv under " + d)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) < 4:
    sys.exit("anchor kernel not found often enough")
i0, i1 = idx[-3], idx[-2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    q = r.get("Queue_Id", "")
    print("%-44s %8.1f -> %8.1f  (%5.1f)  q=%s" % (n, s, e, e - s, q))
