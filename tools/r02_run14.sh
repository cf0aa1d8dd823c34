#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run14; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_q3.json 2> $O/bench_q3.err; tail -c 1200 $O/bench_q3.json | head -c 700; echo
timeout 600 python bench.py --quality 4 --steps 20 --warmup 5 --cpu-seconds 10 > $O/bench_q4.json 2> $O/bench_q4.err
timeout 600 python bench.py --quality 6 --steps 5 --warmup 2 --batch 2 --no-cpu-baseline > $O/bench_q6.json 2> $O/bench_q6.err
timeout 600 python bench.py --size 16384 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_16384.json 2> $O/bench_16384.err
timeout 900 bash tools/profile.sh r02k_q3 > $O/prof_q3.log 2>&1
timeout 900 bash tools/profile.sh r02k_q4 --quality 4 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $O/prof_q4.log 2>&1
( time timeout 1700 python -m pytest tests -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
