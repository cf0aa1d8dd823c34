#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run15; rm -rf $O; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_shard.py tests/test_bands.py -q -m gpu -x -k "batch or bench or shard or band" ) > $O/pytest_quick.log 2>&1; echo "rc=$?" >> $O/pytest_quick.log; tail -4 $O/pytest_quick.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q3_quick.json 2> $O/bench_q3_quick.err; tail -c 600 $O/bench_q3_quick.json; echo
timeout 600 python tools/bench_serving.py 1920 1080 6 3 > $O/serving_q6_coupled.txt 2>&1; tail -9 $O/serving_q6_coupled.txt
QS_HIP_NO_COUPLE=1 timeout 600 python tools/bench_serving.py 1920 1080 6 3 > $O/serving_q6_nocouple.txt 2>&1; tail -9 $O/serving_q6_nocouple.txt
timeout 600 python tools/bench_serving.py 1920 1080 5 3 > $O/serving_q5_coupled.txt 2>&1; tail -5 $O/serving_q5_coupled.txt
( time timeout 1700 python -m pytest tests -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
