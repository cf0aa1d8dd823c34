#!/bin/bash
# round-2 GPU session 26: the 20,000-trial campaign at HEAD (joint predictor as a set launch inside coupled groups)
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run26; rm -rf $O; mkdir -p $O
C=build/fuzz/fuzz_s9.jsonl
( time timeout 900 python tools/fuzz_gpu.py run $C ) > $O/fuzz_head.txt 2>&1; tail -5 $O/fuzz_head.txt | grep -v amdgpu
