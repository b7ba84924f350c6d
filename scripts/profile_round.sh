#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines, rocprofv3 kernel stats of the same command, and the
# separate --pmc passes for HBM traffic (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950).
#   usage: bash scripts/profile_round.sh r01 [bench|pmc|gemm|mfma|all]   (stages, so that one gpurun call stays inside its limit)
set -o pipefail
R=${1:-r01}
STAGE=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R; mkdir -p $O
if [ $STAGE = all ] || [ $STAGE = bench ]; then
python3 bench.py --steps 200 --warmup 20 > $O/bench_ecoli.json 2> $O/bench_ecoli.err || { tail -5 $O/bench_ecoli.err; exit 1; }
python3 bench.py --prec bf16x3 --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 > $O/bench_ecoli_bf16x3.json 2> $O/bench_ecoli_bf16x3.err
python3 scripts/loader_bench.py > $O/loader_bench.txt 2>&1
python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 > $O/bench_block1m.json 2> $O/bench_block1m.err || { tail -5 $O/bench_block1m.err; exit 1; }
python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 --prec f32 > $O/bench_block1m_f32.json 2> $O/bench_block1m_f32.err
python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 --prec bf16x3 > $O/bench_block1m_bf16x3.json 2> $O/bench_block1m_bf16x3.err
python3 bench.py --workload powerlaw --steps 20 --warmup 3 --cpu-seconds 0 > $O/bench_powerlaw.json 2> $O/bench_powerlaw.err
python3 bench.py --model generalgnn --steps 50 --warmup 5 > $O/bench_generalgnn.json 2> $O/bench_generalgnn.err
python3 bench.py --model generalgnn --prec bf16x3 --steps 50 --warmup 5 --cpu-seconds 0 > $O/bench_generalgnn_bf16x3.json 2> $O/bench_generalgnn_bf16x3.err
python3 bench.py --model generalgnn --prec bf16 --steps 50 --warmup 5 --cpu-seconds 0 > $O/bench_generalgnn_bf16.json 2> $O/bench_generalgnn_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ecoli -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-generalgnn > $O/trace_ecoli.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_block1m -- python3 bench.py --workload block1m --steps 20 --warmup 3 --cpu-seconds 0 > $O/trace_block1m.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_powerlaw -- python3 bench.py --workload powerlaw --steps 20 --warmup 3 --cpu-seconds 0 > $O/trace_powerlaw.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_generalgnn -- python3 bench.py --model generalgnn --prec bf16x3 --steps 50 --warmup 5 --cpu-seconds 0 > $O/trace_generalgnn.log 2>&1
fi
if [ $STAGE = all ] || [ $STAGE = pmc ]; then
for w in ecoli block1m powerlaw; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 scripts/spmm_bench.py --workload $w --rounds 1 --iters 5 --slabs 0 > $O/pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write_$w -- python3 scripts/spmm_bench.py --workload $w --rounds 1 --iters 5 --slabs 0 > $O/pmc_write_$w.log 2>&1
done
fi
if [ $STAGE = all ] || [ $STAGE = gemm ]; then
# MFMA utilisation of the weight GEMMs (north_star: "MFMA utilisation on the GEMM against gfx950 peak"): SQ counters in two
# passes (8 SQ slots per pass), config-3 shapes (N = 1M, 256 x 256), every precision
for p in f32 bf16x3 bf16; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_gemm_a_$p -- python3 scripts/gemm_bench.py --n 1000000 --shapes 256x256 --prec $p --iters 3 > $O/pmc_gemm_a_$p.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/pmc_gemm_b_$p -- python3 scripts/gemm_bench.py --n 1000000 --shapes 256x256 --prec $p --iters 3 > $O/pmc_gemm_b_$p.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gemm_$p -- python3 scripts/gemm_bench.py --n 1000000 --shapes 256x256 --prec $p --iters 3 > $O/trace_gemm_$p.log 2>&1
  python3 scripts/gemm_bench.py --n 1000000 --shapes 256x256 --prec $p --iters 10 > $O/gemm_bench_$p.txt 2>&1
done
python3 scripts/mfma_util.py $O > $O/gemm_mfma_util.txt 2>&1
fi
if [ $STAGE = all ] || [ $STAGE = mfma ]; then
# MFMA utilisation of the GEMM kernels INSIDE the steps (the panel GEMMs of GeneralGNN, the bf16-storage streaming GEMMs of config 3)
for w in "generalgnn:--model generalgnn --prec bf16x3 --steps 30 --warmup 5" "block1m:--workload block1m --steps 10 --warmup 3"; do
  tag=${w%%:*}; args="${w#*:} --cpu-seconds 0 --no-config3 --no-generalgnn"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_step_a_$tag -- python3 bench.py $args > $O/pmc_step_a_$tag.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_step_b_$tag -- python3 bench.py $args > $O/pmc_step_b_$tag.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_step_$tag -- python3 bench.py $args > $O/trace_step_$tag.log 2>&1
  python3 scripts/mfma_util_step.py $O/pmc_step_a_$tag $O/pmc_step_b_$tag $O/trace_step_$tag > $O/gemm_mfma_util_step_$tag.txt 2>&1
done
fi
# keep the summaries, drop the per-dispatch traces (large)
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +3M -delete
python3 scripts/summarize_profiles.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
