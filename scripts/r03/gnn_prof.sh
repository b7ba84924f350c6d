#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03l; mkdir -p $O
for p in f32 bf16x3 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$p -- python3 bench.py --model generalgnn --prec $p --steps 50 --warmup 10 --cpu-seconds 0 --no-config3 > $O/t_$p.log 2>&1
  echo "== $p: $(tail -1 $O/t_$p.log | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["value"])')"
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/t_$p/**/*kernel_stats.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    rows.append((float(r["TotalDurationNs"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, n))
tot = sum(r[0] for r in rows)
for t, c, a, n in sorted(rows, reverse=True)[:22]:
    print(f"  {n[:64]:64s} calls {c:6d} ({c/ (rows and 1):.0f}) avg {a:7.1f} us  {100*t/tot:5.1f} %")
PY
done
find $O -name "*kernel_trace.csv" -delete
