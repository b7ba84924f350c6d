#!/bin/bash
# usage: sweep_env.sh VAR "v1 v2 ..." [repeats] [extra bench args]  -- config-2 ms/step per value of one tuning knob
VAR=$1; VALS=$2; REP=${3:-2}; shift 3
for v in $VALS; do
  for i in $(seq $REP); do
    env $VAR=$v python3 bench.py --steps 500 --warmup 50 --cpu-seconds 0 --no-config3 "$@" 2>/dev/null > /tmp/sweep.json
    python3 - "$VAR=$v" <<'PY'
import json, sys
print(sys.argv[1], json.loads(open("/tmp/sweep.json").read().strip().splitlines()[-1])["ms_per_step"])
PY
  done
done
