#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run8; rm -rf $O; mkdir -p $O
ROWS=32,48,64,96,128,160,192,256,384,512,1024
for f in 0 1; do
  QS_HIP_DP=0 timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS > $O/sizes_f${f}_lane.txt 2>&1
  QS_HIP_DP_GROUPS=0 QS_HIP_DP_GROUPS2=100000 timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS > $O/sizes_f${f}_dp2.txt 2>&1
  QS_HIP_DP_GROUPS=100000 timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS > $O/sizes_f${f}_dp4.txt 2>&1
done
grep -h -v amdgpu $O/sizes_f0_*.txt; grep -h -v amdgpu $O/sizes_f1_*.txt
