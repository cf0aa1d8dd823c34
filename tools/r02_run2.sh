#!/bin/bash
# round-2 GPU session 2: integer-multiply / ILP micro-benchmarks, record-prefetch A/B, full GPU
# suite (shim rows path), job-layer profiles of the coupled route.
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run2; rm -rf $O; mkdir -p $O
timeout 300 ./build/ubench_imul > $O/ubench_imul.txt 2>&1; cat $O/ubench_imul.txt
timeout 600 python tools/bench_variants.py 8192 > $O/variants_8192.txt 2>&1; cat $O/variants_8192.txt
timeout 600 python tools/bench_variants.py 1024 > $O/variants_1024.txt 2>&1; cat $O/variants_1024.txt
timeout 600 python tools/bench_variants.py 2880 > $O/variants_2880.txt 2>&1
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_q3.json 2> $O/bench_q3.err; tail -c 300 $O/bench_q3.json
timeout 600 python bench.py --quality 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_q4.json 2> $O/bench_q4.err
timeout 600 python tools/bench_job.py > $O/bench_job.txt 2>&1; cat $O/bench_job.txt
timeout 600 bash tools/prof_job.sh 1920 3 job1080 > $O/prof_job1080.txt 2>&1; cat $O/prof_job1080.txt
timeout 600 bash tools/prof_job.sh 8192 5 job8192 > $O/prof_job8192.txt 2>&1; cat $O/prof_job8192.txt
# scalar-cache behaviour of the recovery kernel (weight rows + records stream through it)
W=/tmp/pmc_sqc; rm -rf $W
( cd /tmp && timeout 300 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVE_CYCLES \
    --output-format csv -d $W -o pmc -- python $R/bench.py --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-verify > $O/pmc_sqc.log 2>&1 )
f=$(find $W -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && (head -1 $f; grep -E "qs_[a-z_]+kernel" $f) > $O/pmc_sqc_counter_collection.csv
python tools/summarize_prof.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
cp $R/gpurun_out/prof_job1080* $R/gpurun_out/prof_job8192* $O/ 2>/dev/null
ls $O
