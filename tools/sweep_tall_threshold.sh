#!/bin/bash
# dev tool: REAL (57-117 atoms, F = 25, 64-wide) at several batch sizes, every layer on the one-graph-per-workgroup kernels
# (HCG_FAMILY_MID=1) vs the wide-layer route forced on (HCG_TALL_MIN_NODES=0): where functional.TALL_MIN_NODES_D64 belongs.
R=${GRAFT_REPO_ROOT:-/root/repo}
for B in ${@:-96 160 256 384}; do
  for mode in mid tall; do
    if [ $mode = mid ]; then export HCG_FAMILY_MID=1; unset HCG_TALL_MIN_NODES; else unset HCG_FAMILY_MID; export HCG_TALL_MIN_NODES=0; fi
    ms=$(python $R/bench.py --config REAL --num-graphs $B --steps 300 --warmup 30 --no-cpu-baseline --no-ragged --sustain 0.5 --distinct-batches 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['config'].get('nodes'))")
    echo "B = $B  $mode: $ms"
  done
done
