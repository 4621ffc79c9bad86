#!/bin/bash
# dev tool: per-kernel averages (rocprofv3) of the default bench for each prebuilt library variant in csrc/_variants/
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp /root/repo/hcatgnet_amd/csrc/_variants/$v.so /root/repo/hcatgnet_amd/csrc/libhcatgnet_hip.so
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python /root/repo/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-graph > /dev/null 2> /tmp/log_$v.txt
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== variant $v: $(grep 'timed eager' /tmp/log_$v.txt)"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['TotalDurationNs']) > 5e5: print("  ", r['Name'].replace('(anonymous namespace)::','')[:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3, 2))
PY
done
