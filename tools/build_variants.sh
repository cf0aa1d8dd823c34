#!/bin/bash
# Build tuning variants of libjpegqs_hip.so into build/variants/ (measurement only).
#   tools/build_variants.sh "name1:-DFLAG=1 -DX=2" "name2:..."
set -e
cd "$(dirname "$0")/../jpeg-quantsmooth_amd/csrc"
OUT=../../build/variants; rm -rf $OUT; mkdir -p $OUT
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function"
for f in qs_tables qs_planes qs_job qs_fused qs_batch qs_shard; do hipcc $HIPFLAGS -x hip -c $f.cpp -o $OUT/$f.o; done
hipcc $HIPFLAGS -c qs_kernels_aux.hip -o $OUT/qs_aux.o
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  hipcc $HIPFLAGS $flags -c qs_kernels.hip -o $OUT/k_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libjpegqs_hip_$name.so $OUT/k_$name.o $OUT/qs_aux.o $OUT/qs_tables.o $OUT/qs_planes.o $OUT/qs_job.o $OUT/qs_fused.o $OUT/qs_batch.o $OUT/qs_shard.o
done
rm -f $OUT/*.o; ls $OUT
