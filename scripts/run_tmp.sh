#!/bin/bash
# scratch: whatever the current measurement needs (see scripts/README.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 2000 --warmup 50 --cpu-seconds 0 --no-config3
