#!/bin/bash
# round-2 GPU session 19: the 20,000-trial campaign again, now that coupled YCbCr jobs of a batch advance in groups
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run19; rm -rf $O; mkdir -p $O
C=build/fuzz/fuzz_s9.jsonl
run() { name=$1; shift; ( time env "$@" timeout 900 python tools/fuzz_gpu.py run $C ) > $O/fuzz_$name.txt 2>&1; tail -4 $O/fuzz_$name.txt | grep -v amdgpu; }
run coupled_default X=1
run coupled_biggroups QS_HIP_COUPLE_BLOCKS=100000000 QS_HIP_COUPLE_SLOTS=1
