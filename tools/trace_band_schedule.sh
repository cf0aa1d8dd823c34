#!/bin/bash
# kernel timeline of tools/bench_band_schedule.py (one GPU, emulated middle band)
#   tools/trace_band_schedule.sh [world]
set -u
WORLD=${1:-8}
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/trace_band; rm -rf $OUT; mkdir -p $OUT
W=/tmp/trace_band; rm -rf $W; mkdir -p $W
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $W -o t -- python $REPO/tools/bench_band_schedule.py --world $WORLD --steps 6 > $OUT/run.log 2>&1
f=$(find $W -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "qs_" in r["Kernel_Name"] or "copy" in r["Kernel_Name"].lower()]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 kernels of each half (simple first, overlapped second)
def show(rs):
    t0 = int(rs[0]["Start_Timestamp"])
    for r in rs:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f'{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get("Queue_Id","?")} grid={r.get("Grid_Size","?")} {r["Kernel_Name"][:60]}')
half = len(rows) // 2
print("# simple schedule (tail)"); show(rows[half - 30:half])
print("# overlapped schedule (tail)"); show(rows[-45:])
PY
cat $OUT/timeline.txt
