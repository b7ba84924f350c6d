#!/bin/bash
# A variant build of libgcnx next to the product library (A/B measurements through GCNX_LIB):
#   scripts/build_variant.sh NAME "-DFLAG ..." [file ...]   -> scripts/variants/libgcnx_NAME.so
# Only the listed .hip files (default: all) are compiled with the extra flags; the rest come from csrc/build/.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; shift 2
B=/tmp/gcnx_variant_$NAME; mkdir -p $B "$ROOT/scripts/variants"
cd "$ROOT/gcn-string_amd/csrc"
make -s >/dev/null
cp build/*.o $B/
FILES=${@:-$(ls *.hip | sed 's/\.hip$//')}
for f in $FILES; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -I/opt/rocm/include -Wno-unused-function \
    -fvisibility=hidden -DGCNX_BUILD $FLAGS -c $f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o -o "$ROOT/scripts/variants/libgcnx_$NAME.so" -ldl
echo "built scripts/variants/libgcnx_$NAME.so"
