#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run27; rm -rf $O; mkdir -p $O
export SERVING_CONFIGS=1x16,1x32,1x64,2x16,4x16
for blocks in 204800 250000 300000 350000; do
  echo "## QS_HIP_COUPLE_BLOCKS=$blocks" >> $O/sweep_q6.txt
  QS_HIP_COUPLE_BLOCKS=$blocks timeout 300 python tools/bench_serving.py 1920 1080 6 3 2>&1 | grep threads >> $O/sweep_q6.txt
done
cut -c1-100 $O/sweep_q6.txt
