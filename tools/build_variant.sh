#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags>"  ->  variants/<name>/libsphhip.so
# A/B builds of the kernels (selected at run time with SPHHIP_LIB=variants/<name>/libsphhip.so).
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/variants/$NAME
mkdir -p "$OUT"
cd "$ROOT/sph_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-result $EXTRA"
for f in ctx grid pairs selftest; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o "$OUT/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libsphhip.so" "$OUT"/ctx.o "$OUT"/grid.o "$OUT"/pairs.o "$OUT"/selftest.o
rm -f "$OUT"/*.o
echo "built $OUT/libsphhip.so"
