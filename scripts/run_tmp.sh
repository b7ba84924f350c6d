#!/bin/bash
# scratch command file for gpurun experiments (gpurun ships the tree, not the shell history); safe to overwrite
python scripts/fused_bench.py
