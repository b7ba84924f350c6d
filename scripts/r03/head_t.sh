#!/bin/bash
# head kernel after the unrolled slab reduction: tests, then its time inside the config-3 step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/head_t; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "head or softmax or gcn2 or golden" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc = 0 ] || exit $rc
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 > $O/t.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/t.log | head -1
python3 scripts/kstats.py $O/t head
find $O -name "*kernel_trace.csv" -delete
