#!/bin/bash
# round-2 GPU session 7: full suite at HEAD, job phase trace, headline bench + profile of HEAD
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run7; rm -rf $O; mkdir -p $O
( time timeout 1700 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
python tools/trace_job.py > $O/trace_job.txt 2>&1; grep -E "rep [123]" $O/trace_job.txt
timeout 600 python tools/bench_job.py > $O/bench_job.txt 2>&1; grep -v amdgpu $O/bench_job.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_q3.json 2> $O/bench_q3.err; tail -c 300 $O/bench_q3.json
timeout 900 bash tools/profile.sh r02g_q3 > $O/prof_q3.log 2>&1
timeout 900 bash tools/profile.sh r02g_q4 --quality 4 --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-verify > $O/prof_q4.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
