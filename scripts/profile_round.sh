#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines, rocprofv3 kernel stats of the same command, and the
# separate --pmc passes for HBM traffic (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950).
#   usage: bash scripts/profile_round.sh r01
set -o pipefail
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R; mkdir -p $O
python3 bench.py --steps 200 --warmup 20 > $O/bench_ecoli.json 2> $O/bench_ecoli.err || { tail -5 $O/bench_ecoli.err; exit 1; }
python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 > $O/bench_block1m.json 2> $O/bench_block1m.err || { tail -5 $O/bench_block1m.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ecoli -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 > $O/trace_ecoli.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_block1m -- python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 > $O/trace_block1m.log 2>&1
for w in ecoli block1m; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 scripts/spmm_bench.py --workload $w --rounds 1 --iters 5 --slabs 0 > $O/pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write_$w -- python3 scripts/spmm_bench.py --workload $w --rounds 1 --iters 5 --slabs 0 > $O/pmc_write_$w.log 2>&1
done
# keep the summaries, drop the per-dispatch traces (large)
find $O -name "*kernel_trace.csv" -delete
python3 scripts/summarize_profiles.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
