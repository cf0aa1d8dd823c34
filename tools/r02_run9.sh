#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run9; rm -rf $O; mkdir -p $O
timeout 900 python tools/bench_serving.py 1920 1080 6 3 > $O/serving_q6.txt 2>&1; grep -v amdgpu $O/serving_q6.txt
timeout 900 python tools/bench_serving.py 1920 1080 3 3 > $O/serving_q3.txt 2>&1; grep -v amdgpu $O/serving_q3.txt
( time timeout 1700 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
