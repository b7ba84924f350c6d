#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 scripts/r03/host_loop_prof.py > gpurun_out/host_loop_prof.txt 2>&1; cat gpurun_out/host_loop_prof.txt
