#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory into a short text summary (committed under profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"{row.get('Name','?')[:90]:90s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} "
                  f"min_ns={row.get('MinNs')} max_ns={row.get('MaxNs')} pct={row.get('Percentage')}")
print("== per-kernel durations from the trace (ns) ==")
for f in find("trace/**/*kernel_trace.csv"):
    d = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            d[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in d.items():
        v.sort()
        print(f"{k[:90]:90s} n={len(v)} mean={sum(v)/len(v):.0f} median={v[len(v)//2]} min={v[0]} max={v[-1]}")
print("== PMC counters: mean per dispatch, per kernel ==")
for f in find("pmc_*/**/*counter_collection.csv"):
    agg = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        if "step_kernel" not in k and "reset" not in k and "rollout" not in k:
            continue
        for cn, v in cs.items():
            print(f"{k[:70]:70s} {cn:28s} n={len(v)} mean={sum(v)/len(v):.6g}")
