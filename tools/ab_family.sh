#!/bin/bash
# dev tool: per-kernel averages of one bench config with the layers forced onto the one-graph-per-workgroup kernels
# (HCG_FAMILY_MID=1) next to the default routing.  usage: tools/ab_family.sh REAL
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for fam in 0 1; do
  rm -rf /tmp/prof_fam$fam
  export HCG_FAMILY_MID=$fam
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fam$fam -- python $R/bench.py --config $1 --steps 100 --warmup 20 \
    --no-cpu-baseline --no-ragged --no-graph --sustain 0.5 > /tmp/fam$fam.json 2> /tmp/fam$fam.log || { echo "family $fam FAILED"; tail -5 /tmp/fam$fam.log; exit 1; }
  echo "== HCG_FAMILY_MID=$fam  $(python -c "import json;d=json.load(open('/tmp/fam$fam.json'));print(d['ms_per_step'],'ms/step')")"
  python - "$(find /tmp/prof_fam$fam -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['TotalDurationNs']) > 5e5: print("  ", r['Name'].replace('(anonymous namespace)::','')[:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3, 2))
PY
done
