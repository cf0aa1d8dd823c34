#!/bin/bash
# round-2 GPU session 1: tests, headline bench (q3 + q4), profiles of HEAD for both kernel
# instantiations, wave-budget variants with SQ counters, colour bench.  Run through gpurun.
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run1; rm -rf $O; mkdir -p $O
nproc > $O/host.txt; lscpu | head -25 >> $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>&1
python -c "import os; print('affinity', len(os.sched_getaffinity(0)))" >> $O/host.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu -s ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_q3.json 2> $O/bench_q3.err; tail -c 600 $O/bench_q3.json
timeout 600 python bench.py --quality 4 --steps 20 --warmup 5 --cpu-seconds 12 > $O/bench_q4.json 2> $O/bench_q4.err
timeout 600 python bench.py --quality 6 --steps 5 --warmup 2 --batch 2 --cpu-seconds 8 > $O/bench_q6.json 2> $O/bench_q6.err
timeout 600 python bench.py --size 16384 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_16384.json 2> $O/bench_16384.err
timeout 900 bash tools/profile.sh r02a_q3 > $O/prof_q3.log 2>&1
timeout 900 bash tools/profile.sh r02a_q4 --quality 4 --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-verify > $O/prof_q4.log 2>&1
# wave budgets 2 / 3 / 4: timing A/B in one process, then SQ counters per variant
timeout 600 python tools/bench_variants.py 8192 > $O/variants_8192.txt 2>&1
for v in occ2 w3 w4; do
  W=/tmp/pmc_$v; rm -rf $W
  ( cd /tmp && QS_HIP_LIB=$R/build/variants/libjpegqs_hip_$v.so timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --output-format csv -d $W -o pmc -- python $R/bench.py --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-verify > $O/pmc_$v.log 2>&1 )
  f=$(find $W -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E "qs_[a-z_]+kernel" $f) > $O/pmc_${v}_counter_collection.csv
done
python tools/summarize_prof.py $O > $O/variants_pmc_summary.txt 2>&1
ls -la $O
