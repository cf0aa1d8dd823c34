#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run4; rm -rf $O; mkdir -p $O
for sz in 8192 2880 1024; do timeout 600 python tools/bench_variants.py $sz > $O/variants_$sz.txt 2>&1; grep -v amdgpu $O/variants_$sz.txt; done
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_q3.json 2> $O/bench_q3.err; tail -c 400 $O/bench_q3.json
timeout 600 python bench.py --quality 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_q4.json 2> $O/bench_q4.err; tail -c 400 $O/bench_q4.json
