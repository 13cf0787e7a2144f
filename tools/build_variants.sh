#!/bin/bash
# developer aid: build libmalio_hip variants that differ in the -D switches of ONE source file
#   tools/build_variants.sh csrc/measure.hip base:"" cw:"-DKS_CW" ...
# -> ma-lio_amd/variants/<name>.so (run them with tools/ab_quick.sh on the GPU box)
set -e
cd "$(dirname "$0")/../ma-lio_amd"
SRC=$1; shift
make -j8 libmalio_hip.so > /dev/null
mkdir -p variants build/variants
find variants -name "*.so" ! -name poison.so -delete
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result -Ibuild"
[ "$SRC" = csrc/measure.hip ] && FLAGS="$FLAGS -fno-slp-vectorize"  # (as the Makefile builds it)
OBJS=$(ls build/csrc/*.o build/host/*.o | grep -v "build/${SRC}.o")
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $SRC -o build/variants/$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so $OBJS build/variants/$name.o -lrt -lpthread -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls -la variants
