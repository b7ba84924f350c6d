#!/bin/bash
# where the arrays land: the same SpMM with 0 / 4 / 12 GiB allocated first, and bench.py's own block1m reading
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for h in "--hog 0" "--hog 4" "--hog 12" "--hog-after-csr 4" "--hog 0"; do
  echo "== $h" >> gpurun_out/place.txt
  timeout -k 10 300 python3 scripts/spmm_bench.py --workload block1m --rounds 2 --iters 20 --slabs 0 $h 2>&1 | grep "round" >> gpurun_out/place.txt || exit 1
done
timeout -k 10 300 python3 bench.py --workload block1m --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/place_bench.json 2>gpurun_out/place_bench.err && python3 -c "
import json
for l in open('gpurun_out/place_bench.json'):
    if l.startswith('{'):
        r=json.loads(l); print('bench block1m roofline', r['roofline']['avg_launch_us'], r['roofline']['frac'])
" >> gpurun_out/place.txt
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-generalgnn > gpurun_out/place_bench2.json 2>gpurun_out/place_bench2.err && python3 -c "
import json
for l in open('gpurun_out/place_bench2.json'):
    if l.startswith('{'):
        r=json.loads(l); print('bench ecoli roofline_config3', r['roofline_config3']['avg_launch_us'], r['roofline_config3']['frac'])
" >> gpurun_out/place.txt
cat gpurun_out/place.txt
