#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run21; rm -rf $O; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch" ) > $O/pytest_batch.log 2>&1; echo "rc=$?" >> $O/pytest_batch.log; tail -3 $O/pytest_batch.log
export SERVING_CONFIGS=1x16,1x32,1x64,2x16,4x16
for blocks in 204800 409600 819200; do for slots in 2 3; do
  echo "## QS_HIP_COUPLE_BLOCKS=$blocks QS_HIP_COUPLE_SLOTS=$slots" >> $O/sweep_q6.txt
  QS_HIP_COUPLE_BLOCKS=$blocks QS_HIP_COUPLE_SLOTS=$slots timeout 300 python tools/bench_serving.py 1920 1080 6 3 2>&1 | grep threads >> $O/sweep_q6.txt
done; done
cut -c1-100 $O/sweep_q6.txt
