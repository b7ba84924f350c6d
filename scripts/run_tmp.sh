#!/bin/bash
GCNX_DW_DIRECT=1 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_backward" 2>&1 | tail -2
python scripts/fused_bench.py | grep dw2
for sl in 16 32 64; do for d in 0 3; do echo "slices $sl dbg $d"; GCNX_DW_SLICES=$sl GCNX_DW_DBG=$d GCNX_DW_DIRECT=1 python scripts/fused_bench.py | grep dw2; done; done
