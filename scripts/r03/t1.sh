#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/t1
timeout -k 10 600 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "bf16_storage or rccl or world_size_2" > gpurun_out/t1/tests.log 2>&1; rc=$?; tail -25 gpurun_out/t1/tests.log; exit $rc
