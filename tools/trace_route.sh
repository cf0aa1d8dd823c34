#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the product route on one device (run on the GPU box):
#   tools/trace_route.sh <tag> [bench_product_route.py args]  ->  gpurun_out/<tag>/route_timeline.txt
set -u
export TMPDIR=/tmp
R=$PWD; TAG=${1:?tag}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
W=/tmp/route_$TAG; rm -rf $W; mkdir -p $W
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $W -o rt -- python $R/tools/bench_product_route.py --devices 0 --reps 2 "$@" > $O/route_run.log 2>&1
python $R/tools/route_timeline.py $W > $O/route_timeline.txt 2>&1
tail -3 $O/route_run.log | cut -c1-600
cat $O/route_timeline.txt
