cd $GRAFT_REPO_ROOT
for v in "" abl1 abl2 abl3; do
  if [ -n "$v" ]; then export GCNX_LIB=$GRAFT_REPO_ROOT/scripts/variants/lib_$v.so; fi
  echo "== variant ${v:-base}"
  for p in bf16x3 bf16; do timeout -k 10 120 python scripts/gemm_bench.py --n 1000000 --shapes 256x256 --prec $p --iters 10; done
done 2>&1
