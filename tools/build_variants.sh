#!/bin/bash
# Build tuning variants of libjpegqs_hip.so into build/variants/ (measurement only).
#   tools/build_variants.sh "name1:-DFLAG=1 -DX=2" "name2:..."
# A spec "name:..." is built from the SHIPPED kernel source (csrc/qs_kernels.hip) when its flags start with "@ship",
# otherwise from the round-3 experiments source with all of its QS_* switches (tools/experiments/qs_kernels_r03.hip).
set -e
cd "$(dirname "$0")/../jpeg-quantsmooth_amd/csrc"
OUT=../../build/variants; rm -rf $OUT; mkdir -p $OUT
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function"
for f in qs_tables qs_planes qs_job qs_fused qs_batch qs_shard; do hipcc $HIPFLAGS -x hip -c $f.cpp -o $OUT/$f.o; done
hipcc $HIPFLAGS -c qs_kernels_aux.hip -o $OUT/qs_aux.o
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  src=../../tools/experiments/qs_kernels_r03.hip
  case "$flags" in "@ship"*) src=qs_kernels.hip; flags=${flags#@ship} ;; esac
  if [ $src = qs_kernels.hip ]; then   # the shipped source gets the shipped build steps (csrc/build_stripped.sh: no-ops between asm statements stripped)
    bash build_stripped.sh $src $OUT/k_$name.o 0 $HIPFLAGS -I. $flags
  else
    hipcc $HIPFLAGS -I. $flags -c $src -o $OUT/k_$name.o
  fi
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libjpegqs_hip_$name.so $OUT/k_$name.o $OUT/qs_aux.o $OUT/qs_tables.o $OUT/qs_planes.o $OUT/qs_job.o $OUT/qs_fused.o $OUT/qs_batch.o $OUT/qs_shard.o
done
rm -f $OUT/*.o; ls $OUT
