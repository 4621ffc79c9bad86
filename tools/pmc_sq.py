#!/usr/bin/env python3
"""Dev tool: per-kernel averages of the SQ counters of one `rocprofv3 --pmc ...` pass (counter_collection.csv under DIR)."""
import csv, glob, os, sys, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if not k.startswith("k_"):
        continue
    print(k)
    for name, v in sorted(c.items()):
        print(f"    {name:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")
