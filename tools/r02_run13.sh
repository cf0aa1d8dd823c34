#!/bin/bash
# round-2 GPU session 13: the long randomised campaign (20,000 trials, expected hashes written by the
# oracle on the build box) through every form of pass B and every job route + the suite at HEAD
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run13; rm -rf $O; mkdir -p $O
C=build/fuzz/fuzz_s9.jsonl
run() { name=$1; shift; ( time env "$@" timeout 900 python tools/fuzz_gpu.py run $C ) > $O/fuzz_$name.txt 2>&1; tail -4 $O/fuzz_$name.txt | grep -v amdgpu; }
run default X=1
run lane QS_HIP_DP=0
run dp4 QS_HIP_DP_GROUPS=100000
run dp2 QS_HIP_DP_GROUPS=0 QS_HIP_DP_GROUPS2=100000
run sharded QS_HIP_DEVICES=0,0,0 QS_HIP_SHARD_MIN_BLOCKS=1
run banded QS_HIP_SPLIT_BLOCKS=60 QS_HIP_BAND_BLOCKS=40
run nofuse QS_HIP_NO_FUSE=1
( time timeout 1700 python -m pytest tests -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
