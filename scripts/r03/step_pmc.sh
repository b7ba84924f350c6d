#!/bin/bash
# HBM traffic per kernel of the config-3 step (bf16 operands, bf16 storage): separate FETCH_SIZE / WRITE_SIZE passes
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/step_pmc; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --workload block1m --steps 3 --warmup 1 --burn-in-ms 0 --cpu-seconds 0 --spmm-iters 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --workload block1m --steps 3 --warmup 1 --burn-in-ms 0 --cpu-seconds 0 --spmm-iters 1 > $O/write.log 2>&1
python3 scripts/pmc_by_kernel.py $O/fetch > $O/fetch_by_kernel.txt
python3 scripts/pmc_by_kernel.py $O/write > $O/write_by_kernel.txt
python3 - <<'PY'
import re
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches\s+(\d+)\s+mean\s+([\d.]+)", l)
        if m: d[m.group(1).strip()] = (int(m.group(3)), float(m.group(4)))
    return d
f, w = load("gpurun_out/step_pmc/fetch_by_kernel.txt"), load("gpurun_out/step_pmc/write_by_kernel.txt")
print(f"{'kernel':62s} {'disp':>5s} {'FETCH x2 MB':>12s} {'WRITE MB':>10s}   (KiB counters; FETCH_SIZE doubled per the guide's gfx950 note)")
for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0))[1] + w.get(k, (0, 0))[1])):
    n = f.get(k, w.get(k))[0]
    print(f"{k:62s} {n:5d} {2 * f.get(k, (0, 0))[1] * 1024 / 1e6:12.1f} {w.get(k, (0, 0))[1] * 1024 / 1e6:10.1f}")
PY
find $O -name "*counter_collection.csv" -size +3M -delete
