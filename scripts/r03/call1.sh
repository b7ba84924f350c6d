#!/bin/bash
# r03 call 1: HBM mixed-traffic microbench, baseline SpMM numbers on this box, phase ablations of the tile tiers / pipe kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03a; mkdir -p $O
scripts/micro/bin/hbm_mix > $O/hbm_mix.txt 2>&1 || { tail -5 $O/hbm_mix.txt; exit 1; }
cat $O/hbm_mix.txt
python3 scripts/spmm_bench.py --workload block1m --iters 20 --rounds 2 --slabs 0,pipe > $O/spmm_base.txt 2>&1 || { tail -5 $O/spmm_base.txt; exit 1; }
cat $O/spmm_base.txt
export GCNX_LIB=$GRAFT_REPO_ROOT/scripts/variants/libgcnx_tuning.so
for d in 0 1 2 4 8 3 6 10 14 15; do
  GCNX_SPMM_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_$d -- python3 scripts/spmm_bench.py --workload block1m --iters 10 --rounds 1 --slabs 0 > $O/abl_$d.log 2>&1
  echo "tiers dbg=$d: $(python3 scripts/kstats.py $O/abl_$d spmm)" | tee -a $O/ablation.txt
done
for d in 0 2 10; do
  GCNX_SPMM_DBG=$d GCNX_SPMM_STAMPS=1 python3 scripts/spmm_bench.py --workload block1m --iters 5 --rounds 1 --slabs pipe > $O/pipe_$d.log 2>&1
  echo "pipe dbg=$d:" >> $O/ablation.txt; grep -h "pipe stamps\|round" $O/pipe_$d.log | tail -6 >> $O/ablation.txt
done
find $O -name "*kernel_trace.csv" -delete
cat $O/ablation.txt
