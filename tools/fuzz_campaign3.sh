O=gpurun_out/r05_fuzz2; mkdir -p $O
F=build/fuzz_r05_20k.jsonl
run() { name=$1; shift; ( time env "$@" python tools/fuzz_gpu.py run $F ) > $O/fuzz_$name.txt 2>&1; grep "^fuzz:" $O/fuzz_$name.txt; }
run default X=1
run lane QS_HIP_DP=0
run dp4 QS_HIP_DP_GROUPS=100000
