#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv (newest under the given dir) as a compact table; per-step
totals when --steps is given.  Dev tool."""
import csv, glob, os, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
tot = 0.0
for r in csv.DictReader(open(f)):
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    per = f" per-step {float(r['TotalDurationNs'])/1e3/steps:7.1f}us" if steps else ""
    tot += float(r["TotalDurationNs"])
    if float(r["Percentage"]) > 0.3:
        print(f"{name[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f}us {r['Percentage']:>6s}%{per}")
if steps: print(f"total GPU time per step: {tot/1e3/steps:.1f} us")
