#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/fused_bench.py 2>&1 | grep dw2
python3 scripts/fused_bench.py 2>&1 | grep dw2
