#!/bin/bash
# kernel-trace stats of the job layer on a colour --quality 6 job (all kernels appear)
#   tools/prof_job.sh [size=4096] [niter=5] [tag=job]
SIZE=${1:-4096}; NITER=${2:-5}; TAG=${3:-job}
export TMPDIR=/tmp; REPO=$PWD; W=/tmp/prof_$TAG; rm -rf $W; mkdir -p $W gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W -o job -- python - <<PY > $REPO/gpurun_out/prof_$TAG.log 2>&1
import sys; sys.path.insert(0, "$REPO")
import jpegqs_pkg
pkg = jpegqs_pkg.load(); hip = pkg.HipQS(); synth = pkg.synth
W_, H_ = ($SIZE, $SIZE * 9 // 16) if $SIZE == 1920 else ($SIZE, $SIZE)
j = synth.synth_ycc(W_, H_, 2, 2, 50); kw = dict(hsamp=j["hsamp"], vsamp=j["vsamp"], colorspace=3, image_size=(W_, H_))
for rep in range(5): hip.do_quantsmooth(j["coefs"], j["quants"], 7, $NITER, **kw)
PY
f=$(find $W -name "*kernel_stats.csv" | head -1); cp $f $REPO/gpurun_out/prof_${TAG}_kernel_stats.csv; python - <<PY
import csv
for r in csv.DictReader(open("$REPO/gpurun_out/prof_${TAG}_kernel_stats.csv")):
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
PY
