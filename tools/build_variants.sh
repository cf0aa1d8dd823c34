#!/bin/bash
# Build tuning variants of libjpegqs_hip.so into build/variants/ (measurement only).
set -e
cd "$(dirname "$0")/../jpeg-quantsmooth_amd/csrc"
OUT=../../build/variants; mkdir -p $OUT
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function"
hipcc $HIPFLAGS -x hip -c qs_host.cpp -o $OUT/qs_host.o
for pin in 1 0; do for mw in 1 2 3 4; do for slp in slp noslp; do
  extra=""; [ $slp = noslp ] && extra="-fno-slp-vectorize"
  name=pin${pin}_mw${mw}_${slp}
  hipcc $HIPFLAGS -DQS_PIN_DIFFS=$pin -DQS_SMOOTH_MIN_WAVES=$mw $extra -c qs_kernels.hip -o $OUT/k_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libjpegqs_hip_$name.so $OUT/k_$name.o $OUT/qs_host.o
done; done; done
rm -f $OUT/*.o; ls $OUT
