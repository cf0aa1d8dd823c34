#!/bin/bash
# the 20,000-trial corpus (seed 77, tools/fuzz_gpu.py gen on the build box) replayed in every form of pass B and on the sharded / general routes
O=gpurun_out/${FUZZ_TAG:-r04_fuzz}; mkdir -p $O
F=build/fuzz_r03_20k.jsonl
run() { name=$1; shift; ( time env "$@" python tools/fuzz_gpu.py run $F ) > $O/fuzz_$name.txt 2>&1; tail -4 $O/fuzz_$name.txt | head -2; }
run default X=1
run lane QS_HIP_DP=0
run dp4 QS_HIP_DP_GROUPS=100000
run dp2 QS_HIP_DP_GROUPS=0 QS_HIP_DP_GROUPS2=100000
run sharded QS_HIP_DEVICES=0,0,0 QS_HIP_SHARD_MIN_BLOCKS=1
run nofuse QS_HIP_NO_FUSE=1
run banded QS_HIP_SPLIT_BLOCKS=60 QS_HIP_BAND_BLOCKS=40
