cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p40
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p40 -- python $GRAFT_REPO_ROOT/bench.py --config REAL40 --steps 200 --warmup 20 --no-cpu-baseline --no-ragged --no-graph --sustain 0.5 > /tmp/r40.json 2> /tmp/r40.log || { tail -5 /tmp/r40.log; exit 1; }
python - "$(find /tmp/p40 -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['TotalDurationNs']) > 1e5: print("  ", r['Name'].replace('(anonymous namespace)::','')[:90].ljust(90), r['Calls'], round(float(r['AverageNs'])/1e3, 2))
PY
python -c "import json;d=json.load(open('/tmp/r40.json'));print(d['ms_per_step'], d.get('launch'), d.get('launch_forms_ms'))"
