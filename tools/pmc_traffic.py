#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: counters are in KiB-units of 64-B
requests; FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced streaming reads -> doubled.
Writes profiles/<tag>_traffic.json.  With a 4th argument SECTION (C3 / C5 / REAL ...) the result becomes that section of a
per-workload file (bench.py reads the section of the config it runs).  Dev tool."""
import csv, glob, json, os, sys, collections
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
def load(d, name):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}
fe, n = load(fetch_dir, "FETCH_SIZE")
wr, _ = load(write_dir, "WRITE_SIZE")
res = {}
for k in sorted(fe, key=lambda k: -(fe[k] + wr.get(k, 0))):
    if not k.startswith("k_"):
        continue
    fetch_b = fe[k] * 1024 * 2          # gfx950: FETCH_SIZE = 1/2 of wide coalesced reads
    write_b = wr.get(k, 0.0) * 1024
    res[k] = {"launches": n[k], "fetch_bytes_corrected": fetch_b, "write_bytes": write_b, "hbm_bytes": fetch_b + write_b,
              "FETCH_SIZE_raw_KiB": fe[k], "WRITE_SIZE_raw_KiB": wr.get(k, 0.0)}
    print(f"{k:45s} n={n[k]:4d} fetch {fetch_b/1e6:8.2f} MB (raw {fe[k]/1e3:7.2f} MKiB) write {write_b/1e6:8.2f} MB total {(fetch_b+write_b)/1e6:8.2f} MB")
if len(sys.argv) > 4:
    allsec = json.load(open(out)) if os.path.isfile(out) else {}
    allsec[sys.argv[4]] = res
    res = allsec
json.dump(res, open(out, "w"), indent=1)
