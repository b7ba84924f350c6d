#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03j; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard8 -- python3 bench.py --workload block1m --steps 30 --warmup 3 --cpu-seconds 0 --no-config3 --spmm-iters 2 --emulate-rank 0 --of 8 > $O/shard8.json 2> $O/shard8.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full -- python3 bench.py --workload block1m --steps 30 --warmup 3 --cpu-seconds 0 --no-config3 --spmm-iters 2 > $O/full.json 2> $O/full.err
python3 - <<PY
import csv, glob
def load(d):
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
    out = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3)
    return out
a, b = load("$O/shard8"), load("$O/full")
steps = 30 + 3 + 5 + 3   # rough: timed + warmup + burn-in varies; use calls ratio instead
print(f"{'kernel':70s} {'calls8':>7s} {'avg8 us':>9s} {'callsF':>7s} {'avgF us':>9s} {'F/8':>8s} {'ratio':>6s}")
tot8 = totF = 0
for k in sorted(set(a) | set(b), key=lambda k: -(b.get(k, (0, 0, 0))[2])):
    ca, aa, ta = a.get(k, (0, 0, 0)); cb, ab, tb = b.get(k, (0, 0, 0))
    print(f"{k[:70]:70s} {ca:7d} {aa:9.1f} {cb:7d} {ab:9.1f} {ab/8:8.1f} {(aa/(ab/8) if ab else 0):6.2f}")
PY
find $O -name "*kernel_trace.csv" -delete
