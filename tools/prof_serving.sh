#!/bin/bash
# rocprofv3 kernel statistics of the batched job layer (tools/bench_serving, 1 thread, batch 32)
#   tools/prof_serving.sh <tag> [flags=0]   -> gpurun_out/prof_<tag>/     (flags 7 = --quality 6)
set -u
TAG=${1:-serving}
FLAGS=${2:-0}
export TMPDIR=/tmp
REPO=$PWD
SERVING_CONFIGS=1x2 python tools/bench_serving.py > /dev/null 2>&1   # builds /tmp/qs_serving/{bench_serving,job.bin}
OUT=$REPO/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
W=/tmp/prof_$TAG; rm -rf $W; mkdir -p $W
cd /tmp
/tmp/qs_serving/bench_serving /tmp/qs_serving/job.bin $FLAGS 3 1 256 32 > $OUT/unprofiled.json 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $W -o s -- /tmp/qs_serving/bench_serving /tmp/qs_serving/job.bin $FLAGS 3 1 256 32 > $OUT/profiled.log 2>&1
for f in $(find $W -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv" -o -name "*domain_stats.csv"); do cp $f $OUT/$(basename $f); done
cat $OUT/unprofiled.json; head -8 $OUT/*kernel_stats.csv; head -5 $OUT/*memory_copy_stats.csv 2>/dev/null
