#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/t1
timeout -k 10 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "learning_rate or golden or fit or loader or world_size_2 or rccl" > gpurun_out/t1/tests.log 2>&1; rc=$?; tail -25 gpurun_out/t1/tests.log; exit $rc
