#!/bin/bash
# full GPU test suite + default bench line (what the driver runs at round end)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03full}; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -3 $O/gputests.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python3 - <<PY
import json
r=json.load(open("$O/bench_default.json"))
print("ms/step", r["ms_per_step"], "graphs/s", r["value"], "roofline", r["roofline"]["frac"], "cfg3", r["roofline_config3"]["frac"], r["roofline_config3"]["avg_launch_us"])
PY
