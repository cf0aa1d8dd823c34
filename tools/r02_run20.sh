#!/bin/bash
# round-2 GPU session 20: suite + bench lines at HEAD (after the coupled-group route and the packed halo exchange)
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02_run20; rm -rf $O; mkdir -p $O
( time timeout 1700 python -m pytest tests -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 600 python bench.py > $O/bench_q3.json 2> $O/bench_q3.err; tail -c 900 $O/bench_q3.json | head -c 500; echo
timeout 600 python bench.py --quality 4 --cpu-seconds 8 > $O/bench_q4.json 2> $O/bench_q4.err
timeout 600 python bench.py --quality 6 --steps 5 --warmup 2 --batch 2 --no-cpu-baseline > $O/bench_q6.json 2> $O/bench_q6.err
python -c "
import json,sys
for n in ('q3','q4','q6'):
    try:
        d=json.loads(open('$O/bench_%s.json'%n).read().strip().splitlines()[-1]); print(n, round(d['value']/1e6,1),'M blocks/s', d['ms_per_step'], d.get('verify_ok'), d.get('cpu_baseline',{}).get('value'))
    except Exception as e: print(n,'ERR',e)
"
