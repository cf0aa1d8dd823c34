#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run11; rm -rf $O; mkdir -p $O
ROWS=8,16,32,48,64,96,128,192,256,512
V=$R/build/variants/libjpegqs_hip_dpnoshare.so
L=$R/jpeg-quantsmooth_amd/libjpegqs_hip.so
for f in 0 1; do
  QS_HIP_DP=0 timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS $L > $O/sizes_f${f}_lane.txt 2>&1
  QS_HIP_DP_GROUPS=100000 timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS $L $V > $O/sizes_f${f}_dp4.txt 2>&1
  QS_HIP_DP_GROUPS=0 QS_HIP_DP_GROUPS2=100000 timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS $L $V > $O/sizes_f${f}_dp2.txt 2>&1
done
grep -h -v amdgpu $O/sizes_f0_*.txt; grep -h -v amdgpu $O/sizes_f1_*.txt
