#!/bin/bash
# dev tool: per-kernel averages (rocprofv3 --kernel-trace --stats) of one bench config for prebuilt library variants
# (hcatgnet_amd/csrc/_variants/<name>.so, loaded through HCG_LIB).  usage: [ABK_ARGS="--no-parity-gate"] tools/ab_kernels.sh "CFG1 CFG2" A B ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cfgs=$1; shift
for v in "$@"; do
  for c in $cfgs; do
    rm -rf /tmp/prof_${v}_$c
    export HCG_LIB=$R/hcatgnet_amd/csrc/_variants/$v.so
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${v}_$c -- python $R/bench.py --config $c --steps 200 --warmup 20 \
      --no-cpu-baseline --no-ragged --no-graph $ABK_ARGS > /tmp/abk_${v}_$c.json 2> /tmp/abk_${v}_$c.log || { echo "$v $c FAILED"; tail -5 /tmp/abk_${v}_$c.log; exit 1; }
    f=$(find /tmp/prof_${v}_$c -name "*kernel_stats.csv" | head -1)
    echo "== variant $v config $c"
    python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['TotalDurationNs']) > 5e5: print("  ", r['Name'].replace('(anonymous namespace)::','')[:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3, 2))
PY
  done
done
