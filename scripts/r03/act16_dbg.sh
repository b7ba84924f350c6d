#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/act16
timeout -k 10 500 python3 scripts/r03/act16_dbg.py > gpurun_out/act16/dbg.txt 2>&1; tail -20 gpurun_out/act16/dbg.txt
