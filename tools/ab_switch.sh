#!/bin/bash
# dev tool: per-kernel averages + ms/step of REAL and C5 with one FusedTrainStep switch on / off, interleaved twice.
# usage: tools/ab_switch.sh [ENV_SWITCH=HCG_NO_POOLBITS] [CONFIGS="REAL C5"]
SW=${1:-HCG_NO_POOLBITS}
CFGS=${2:-"REAL C5"}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for mode in bits nobits; do   # "bits" = default build, "nobits" = the switch set
  for c in $CFGS; do
    if [ $mode = nobits ]; then export $SW=1; else unset $SW; fi
    rm -rf /tmp/prof_${mode}_$c
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${mode}_$c -- python $R/bench.py --config $c --steps 100 --warmup 20 --no-cpu-baseline --no-ragged --no-graph > /tmp/abb_${mode}_$c.json 2> /tmp/abb_${mode}_$c.log || { echo "$mode $c FAILED"; tail -5 /tmp/abb_${mode}_$c.log; exit 1; }
    f=$(find /tmp/prof_${mode}_$c -name "*kernel_stats.csv" | head -1)
    echo "== $mode $c  $(python -c "import json;d=json.loads(open('/tmp/abb_${mode}_$c.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d.get('parity_gate',{}).get('worst'))")"
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:10]:
    print("   ", r["Name"].replace("(anonymous namespace)::", "")[:72], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
  done
done
done
