#!/usr/bin/env python3
"""Timeline of ONE product-route call (qs_hip_do_quantsmooth on an 8192^2 plane, host arrays in and out) from a
rocprofv3 --kernel-trace --memory-copy-trace run of tools/bench_product_route.py: every copy and kernel of the LAST
call with start / end relative to the call's first event, so that what overlaps and what waits can be read off.
    tools/route_timeline.py <rocprof output dir>"""
import csv
import sys
from pathlib import Path

root = Path(sys.argv[1])
ev = []
for f in root.rglob("*.csv"):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames or []
        s = next((c for c in cols if c.lower().startswith("start")), None)
        e = next((c for c in cols if c.lower().startswith("end")), None)
        if not s or not e:
            continue
        name = next((c for c in cols if c in ("Kernel_Name", "Direction", "Name")), None)
        q = next((c for c in cols if c in ("Queue_Id", "Stream_Id")), None)
        nbytes = next((c for c in cols if "byte" in c.lower() or c.lower() == "size"), None)
        for row in rd:
            try:
                ev.append((int(row[s]), int(row[e]), (row.get(name, "?") if name else "?")[:44], row.get(q, "") if q else "",
                           row.get(nbytes, "") if nbytes else ""))
            except (ValueError, TypeError):
                pass
ev.sort()
if not ev:
    sys.exit("no events found under " + str(root))
# calls are separated by gaps > 3 ms of nothing on the device
calls, cur = [], [ev[0]]
for x in ev[1:]:
    if x[0] - max(y[1] for y in cur) > 3_000_000:
        calls.append(cur); cur = []
    cur.append(x)
calls.append(cur)
big = [c for c in calls if sum(1 for x in c if "smooth" in x[2]) >= 3]
call = big[-1] if big else calls[-1]
t0 = call[0][0]
print(f"# {len(calls)} bursts of device activity; the last one with recovery kernels: {len(call)} events, {(max(x[1] for x in call) - t0) / 1e6:.3f} ms from first start to last end")
print(f"# {'start ms':>9s} {'end ms':>9s} {'dur ms':>8s}  {'queue':>6s} {'bytes':>11s}  what")
for s, e, n, q, b in call:
    print(f"  {(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f}  {q:>6s} {b:>11s}  {n}")
busy = {}
for s, e, n, q, b in call:
    k = "H2D" if "HOST_TO_DEVICE" in n.upper() or "H2D" in n.upper() else "D2H" if "DEVICE_TO_HOST" in n.upper() or "D2H" in n.upper() else "smooth" if "smooth" in n else "idct" if "idct" in n else "other"
    busy[k] = busy.get(k, 0) + (e - s)
print("# busy time by kind (ms, overlaps not merged):", {k: round(v / 1e6, 3) for k, v in busy.items()})
