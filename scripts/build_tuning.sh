#!/bin/bash
# A TUNING build of libgcnx (-DGCNX_TUNING: phase-ablation bits, wave stamps) next to the product library:
#   scripts/variants/libgcnx_tuning.so   (git-ignored; use through GCNX_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/gcnx_tuning_build; mkdir -p $B "$ROOT/scripts/variants"
cd "$ROOT/gcn-string_amd/csrc"
for f in runtime graph_prep spmm spmm_bf16 fused gemm gemm_stream gemm_panel reduce bn elementwise head comm; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -I/opt/rocm/include -Wno-unused-function \
    -fvisibility=hidden -DGCNX_BUILD -DGCNX_TUNING -c $f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o -o "$ROOT/scripts/variants/libgcnx_tuning.so" -ldl
echo "built $ROOT/scripts/variants/libgcnx_tuning.so"
