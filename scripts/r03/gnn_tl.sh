#!/bin/bash
# one GeneralGNN step as a kernel timeline (bf16x3), with grid sizes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gnn_tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --model generalgnn --prec bf16x3 --steps 10 --warmup 3 --burn-in-ms 0 > $O/t.log 2>&1
python3 - $O/t <<'PY' > $O/timeline.txt
import csv, glob, os, sys
fs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
i0, i1 = idx[-3] + 1, idx[-2]
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
for r in rows[i0:i1 + 1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
    s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    print("%-46s start %8.1f dur %6.1f gap %5.1f grid %s wg %s" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", ""))))
    prev = e
PY
cat $O/timeline.txt | head -120
find $O -name "*kernel_trace.csv" -delete
