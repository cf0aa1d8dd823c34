#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run17; rm -rf $O; mkdir -p $O
timeout 300 python tools/bench_variants.py 8192 > $O/variants_8192.txt 2>&1; cat $O/variants_8192.txt
timeout 200 python tools/bench_variants.py 2048 > $O/variants_2048.txt 2>&1; cat $O/variants_2048.txt
