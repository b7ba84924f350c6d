#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "spmm or pool_bwd or config3_and_config5" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 scripts/spmm_bench.py --workload powerlaw --iters 20 --rounds 2 --slabs 0 2>&1 | grep round
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 scripts/spmm_bench.py --workload powerlaw --iters 10 --rounds 1 --slabs 0 > $O/tr.log 2>&1
echo "kernels: $(python3 scripts/kstats.py $O/tr spmm)"
python3 bench.py --workload powerlaw --steps 20 --warmup 3 --cpu-seconds 0 > $O/bench_powerlaw.json 2> $O/bench_powerlaw.err || tail -3 $O/bench_powerlaw.err
python3 -c "import json; r=json.load(open('$O/bench_powerlaw.json')); print('powerlaw step', r['ms_per_step'], 'roofline', r['roofline']['frac'], r['roofline']['avg_launch_us'])"
find $O -name "*kernel_trace.csv" -delete
