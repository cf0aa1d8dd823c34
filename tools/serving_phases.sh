#!/bin/bash
# per-phase times (QS_HIP_TRACE) of tools/bench_serving: [threads batch] pairs
python tools/bench_serving.py > /dev/null 2>&1   # builds /tmp/qs_serving/{bench_serving,job.bin}
for tb in "1 1" "4 1" "1 8" "1 16" "2 8"; do
  set -- $tb
  QS_HIP_TRACE=1 /tmp/qs_serving/bench_serving /tmp/qs_serving/job.bin 0 3 $1 64 $2 2> /tmp/qs_serving/trace.txt
  grep "qs_hip trace" /tmp/qs_serving/trace.txt | tail -n 4
done
