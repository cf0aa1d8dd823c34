#!/bin/bash
# round-2 GPU session 6: the diagonal-parallel small-plane kernel -- correctness (suite, both forms
# forced over the corpus) and where it beats the one-block-per-lane kernel
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run6; rm -rf $O; mkdir -p $O
ROWS=1,2,4,8,16,24,32,40,48,64,96,128
for f in 0 1; do
  QS_HIP_DP=0 timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS > $O/sizes_f${f}_lane.txt 2>&1
  QS_HIP_DP_GROUPS6=100000 timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS > $O/sizes_f${f}_dp6.txt 2>&1
  QS_HIP_DP_GROUPS6=0 QS_HIP_DP_GROUPS4=100000 timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS > $O/sizes_f${f}_dp4.txt 2>&1
  timeout 300 python tools/bench_sizes.py --flags $f --rows $ROWS > $O/sizes_f${f}_default.txt 2>&1
done
grep -h -v amdgpu $O/sizes_f0_*.txt
grep -h -v amdgpu $O/sizes_f1_*.txt
( time timeout 1700 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 600 python tools/bench_job.py > $O/bench_job.txt 2>&1; grep -v amdgpu $O/bench_job.txt
QS_HIP_DP=0 timeout 600 python tools/bench_job.py > $O/bench_job_nodp.txt 2>&1; grep -v amdgpu $O/bench_job_nodp.txt
