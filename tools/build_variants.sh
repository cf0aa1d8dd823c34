#!/bin/bash
# Build tuning variants of libjpegqs_hip.so into build/variants/ (measurement only).
set -e
cd "$(dirname "$0")/../jpeg-quantsmooth_amd/csrc"
OUT=../../build/variants; rm -rf $OUT; mkdir -p $OUT
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function"
hipcc $HIPFLAGS -x hip -c qs_host.cpp -o $OUT/qs_host.o
for pipe in 1 0; do for mw in 2 3 4; do
  name=pipe${pipe}_mw${mw}
  hipcc $HIPFLAGS -DQS_SMEM_PIPELINE=$pipe -DQS_SMOOTH_MIN_WAVES=$mw -c qs_kernels.hip -o $OUT/k_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libjpegqs_hip_$name.so $OUT/k_$name.o $OUT/qs_host.o
done; done
rm -f $OUT/*.o; ls $OUT
cd "$(dirname "$0")/../jpeg-quantsmooth_amd/csrc" 2>/dev/null || true
OUT=../../build/variants
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function"
hipcc $HIPFLAGS -x hip -c qs_host.cpp -o $OUT/qs_host.o
for ab in IDCT UPDATE "IDCT -DQS_ABLATE_UPDATE"; do
  name=ablate_$(echo $ab | tr -d ' -' | sed 's/DQS_ABLATE_/_/')
  hipcc $HIPFLAGS -DQS_ABLATE_$ab -c qs_kernels.hip -o $OUT/k_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libjpegqs_hip_$name.so $OUT/k_$name.o $OUT/qs_host.o
done
rm -f $OUT/*.o; ls $OUT
