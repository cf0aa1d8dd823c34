#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run12; rm -rf $O; mkdir -p $O
( time timeout 1700 python -m pytest tests -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 600 python tools/bench_job.py > $O/bench_job.txt 2>&1; grep -v amdgpu $O/bench_job.txt
python tools/trace_job.py > $O/trace_job.txt 2>&1; grep -E "rep [123]" $O/trace_job.txt
timeout 900 python tools/bench_serving.py 1920 1080 3 3 > $O/serving_q3.txt 2>&1; grep -v amdgpu $O/serving_q3.txt | head -8
