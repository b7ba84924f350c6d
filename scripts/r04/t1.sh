#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --workload block1m --emulate-rank 0 --of 8 --steps 100 --warmup 10 --cpu-seconds 0 2>/dev/null | tail -1 | head -c 400
