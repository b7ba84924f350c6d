#!/bin/bash
# A/B of the folded pool backward (GCNX_FOLD) on config 2: alternating bench runs + per-kernel stats of both.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; mkdir -p $O
for i in 1 2; do
  GCNX_FOLD=0 python3 bench.py --steps 500 --warmup 50 --cpu-seconds 0 --no-config3 > $O/f0_$i.json 2>/dev/null
  GCNX_FOLD=1 python3 bench.py --steps 500 --warmup 50 --cpu-seconds 0 --no-config3 > $O/f1_$i.json 2>/dev/null
done
python3 - <<PY
import json
for k in ("f0","f1"):
    print(k, [round(json.loads(open("$O/%s_%d.json"%(k,i)).read().strip().splitlines()[-1])["ms_per_step"],4) for i in (1,2)])
PY
export GCNX_FOLD=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t0 -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 > $O/t0.log 2>&1
export GCNX_FOLD=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t1 -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 > $O/t1.log 2>&1
for t in t0 t1; do echo == $t; python3 scripts/kstats.py $(find $O/$t -name "*kernel_stats.csv" | head -1) | head -16; done
for t in t0 t1; do echo == timeline $t; python3 scripts/ktimeline.py $O/$t; done
find $O -name "*kernel_trace.csv" -delete
