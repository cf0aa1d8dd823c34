#!/bin/bash
# round-2 GPU session 24: rocprofv3 evidence of bench.py at HEAD (q3) + the GPU suite at HEAD
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run24; rm -rf $O; mkdir -p $O
timeout 900 bash tools/profile.sh r02p_q3 > $O/prof_q3.log 2>&1; tail -5 $O/prof_q3.log
cd $R
( time timeout 1700 python -m pytest tests -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
