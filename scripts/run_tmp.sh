#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "20 5" "20 5" "200 20" "2000 50"; do set -- $cfg
python3 bench.py --steps $1 --warmup $2 --cpu-seconds 0 --no-config3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('steps $1 warmup $2:', round(d['ms_per_step'],5), round(d['value']), 'median', round(d['m1_median']['ms_per_step'],5), d['burn_in']['steps'])"
done
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-config3 --burn-in-ms 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('no burn-in:', round(d['ms_per_step'],5), d['burn_in']['steps'])"
python3 bench.py --workload block1m --steps 10 --warmup 3 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('block1m:', round(d['ms_per_step'],4), d['burn_in']['steps'])"
