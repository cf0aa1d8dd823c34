#!/bin/bash
# GPU timeline (kernels + copies) of tools/bench_serving:  tools/trace_serving.sh <threads> <batch> [jobs per thread]
set -u
T=${1:-1}; B=${2:-8}; PER=${3:-32}
export TMPDIR=/tmp
REPO=$PWD
python tools/bench_serving.py > /dev/null 2>&1   # builds /tmp/qs_serving/{bench_serving,job.bin}
OUT=$REPO/gpurun_out/trace_serving; rm -rf $OUT; mkdir -p $OUT
W=/tmp/trace_serving; rm -rf $W; mkdir -p $W
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $W -o t -- /tmp/qs_serving/bench_serving /tmp/qs_serving/job.bin 0 3 $T $PER $B > $OUT/run.log 2>&1
python - $W > $OUT/timeline.txt <<'PY'
import csv, sys, glob
w = sys.argv[1]
ev = []
for f in glob.glob(w + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:28], r.get("Queue_Id", "?")))
for f in glob.glob(w + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?")[:24] + " " + r.get("Size", ""), "-"))
ev.sort()
# the last third of the run
ev = ev[len(ev) * 2 // 3:][:150]
t0 = ev[0][0]
for s, e, n, q in ev:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us q{q} {n}")
PY
cat $OUT/run.log | tail -2; head -130 $OUT/timeline.txt
