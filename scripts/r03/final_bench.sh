#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r03/bench_default.json"):
    if l.startswith("{"):
        r = json.loads(l)
        print("default:", r["metric"], round(r["value"]), r["unit"], r["ms_per_step"], "roofline", round(r["roofline"]["frac"], 3), "config3", round(r["roofline_config3"]["frac"], 3),
              "cold", round(r["cold"]["ms_per_step"], 4), "gnn", {k: round(v["ms_per_step"], 3) for k, v in r["generalgnn"].items() if isinstance(v, dict)},
              "cpu", round(r["cpu_baseline"]["value"]), "steps", r["steps"], "warmup", r["warmup"])
PY
python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-generalgnn > $O/bench_ecoli_c.json 2>/dev/null; grep -o '"ms_per_step": [0-9.]*' $O/bench_ecoli_c.json | head -1
