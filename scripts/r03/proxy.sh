#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python3 scripts/scaling_proxy.py --out gpurun_out/scaling_proxy.json > gpurun_out/scaling_proxy.log 2>&1; tail -8 gpurun_out/scaling_proxy.log
