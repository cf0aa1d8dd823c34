#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run10; rm -rf $O; mkdir -p $O
( time timeout 1700 python -m pytest tests -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 600 python tools/bench_job.py > $O/bench_job.txt 2>&1; grep -v amdgpu $O/bench_job.txt
