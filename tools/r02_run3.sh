#!/bin/bash
# round-2 GPU session 3: record prefetch (fixed) and the 4-waves-per-SIMD variant, A/B + suite
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run3; rm -rf $O; mkdir -p $O
for sz in 8192 2880 1024; do timeout 600 python tools/bench_variants.py $sz > $O/variants_$sz.txt 2>&1; grep -v amdgpu $O/variants_$sz.txt; done
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_q3.json 2> $O/bench_q3.err; tail -c 300 $O/bench_q3.json
QS_HIP_LIB=$R/build/variants/libjpegqs_hip_w4e.so timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_q3_w4e.json 2> $O/bench_q3_w4e.err; tail -c 300 $O/bench_q3_w4e.json
QS_HIP_LIB=$R/build/variants/libjpegqs_hip_w4e.so timeout 600 python bench.py --quality 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_q4_w4e.json 2> $O/bench_q4_w4e.err
W=/tmp/pmc_w4e; rm -rf $W
( cd /tmp && QS_HIP_LIB=$R/build/variants/libjpegqs_hip_w4e.so timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $W -o pmc -- python $R/bench.py --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-verify > $O/pmc_w4e.log 2>&1 )
f=$(find $W -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && (head -1 $f; grep -E "qs_[a-z_]+kernel" $f) > $O/pmc_w4e_counter_collection.csv
python tools/summarize_prof.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
