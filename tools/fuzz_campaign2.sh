#!/bin/bash
# a FRESH 20,000-trial corpus (tools/fuzz_gpu.py gen build/fuzz_r04_20k.jsonl 20000 404, oracle only, on the build box) replayed in
# four forms of pass B (all with the fused epilogue) and on the banded / sharded routes
O=gpurun_out/${FUZZ_TAG:-r04_fuzz2}; mkdir -p $O
F=${FUZZ_FILE:-build/fuzz_r04_20k.jsonl}
run() { name=$1; shift; ( time env "$@" python tools/fuzz_gpu.py run $F ) > $O/fuzz_$name.txt 2>&1; tail -4 $O/fuzz_$name.txt | head -2; }
run default X=1
run lane QS_HIP_DP=0
run dp4 QS_HIP_DP_GROUPS=100000
run dp2 QS_HIP_DP_GROUPS=0 QS_HIP_DP_GROUPS2=100000
run banded QS_HIP_SPLIT_BLOCKS=60 QS_HIP_BAND_BLOCKS=40
