#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + per-dispatch PMC rows) into a
per-kernel summary: mean duration, mean counter values per launch."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])
for f in sorted(out.glob("*kernel_stats.csv")):
    print(f"== {f.name}")
    with open(f) as fh:
        for row in list(csv.DictReader(fh))[:6]:
            print("  {Name:60.60s} calls={Calls:>5} total_ns={TotalDurationNs:>12} avg_ns={AverageNs:>12} pct={Percentage}".format(**row))
for f in sorted(out.glob("*counter_collection.csv")):
    print(f"== {f.name}")
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"][:48]
            if not k.lstrip("void ").startswith("qs_"):
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Scratch_Size"))
    for k, cs in acc.items():
        print(f"  {k}  vgpr/agpr/sgpr/lds/scratch={meta[k]}")
        for c, vals in cs.items():
            print(f"      {c:24s} launches={len(vals):4d} mean={sum(vals) / len(vals):.6g}")
