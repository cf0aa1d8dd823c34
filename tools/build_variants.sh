#!/bin/bash
# Build tuning variants of libjpegqs_hip.so into build/variants/ (measurement only).
#   tools/build_variants.sh "name1:-DFLAG=1 -DX=2" "name2:..."
# A spec "name:..." is built from the SHIPPED kernel source (csrc/qs_kernels.hip) when its flags start with "@ship",
# otherwise from the round-3 experiments source with all of its QS_* switches (csrc/experiments/qs_kernels_r03.hip).
set -e
cd "$(dirname "$0")/../jpeg-quantsmooth_amd/csrc"
OUT=../../build/variants; rm -rf $OUT; mkdir -p $OUT
L=/opt/rocm/lib/llvm/bin
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function"
for f in qs_tables qs_planes qs_job qs_fused qs_batch qs_shard; do hipcc $HIPFLAGS -x hip -c $f.cpp -o $OUT/$f.o; done
hipcc $HIPFLAGS -c qs_kernels_aux.hip -o $OUT/qs_aux.o
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  src=experiments/qs_kernels_r03.hip
  case "$flags" in "@ship"*) src=qs_kernels.hip; flags=${flags#@ship} ;; esac
  if [ $src = qs_kernels.hip ]; then   # the shipped source gets the shipped build steps (csrc/Makefile: no-ops between asm statements stripped)
    hipcc $HIPFLAGS -I. $flags -S --cuda-device-only $src -o $OUT/k.s 2>/dev/null
    python3 strip_asm_nops.py $OUT/k.s $OUT/k_dev.s
    $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $OUT/k_dev.s -o $OUT/k_dev.o
    $L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $OUT/k.co $OUT/k_dev.o
    $L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$OUT/k.co -output=$OUT/k.hipfb
    hipcc $HIPFLAGS -I. $flags --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $OUT/k.hipfb -c $src -o $OUT/k_$name.o
  else
    hipcc $HIPFLAGS -I. $flags -c $src -o $OUT/k_$name.o
  fi
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libjpegqs_hip_$name.so $OUT/k_$name.o $OUT/qs_aux.o $OUT/qs_tables.o $OUT/qs_planes.o $OUT/qs_job.o $OUT/qs_fused.o $OUT/qs_batch.o $OUT/qs_shard.o
done
rm -f $OUT/*.o $OUT/k.s $OUT/k_dev.s $OUT/k.co $OUT/k.hipfb; ls $OUT
