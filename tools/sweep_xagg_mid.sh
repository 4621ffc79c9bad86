#!/bin/bash
# dev tool: REAL at small / mid batch sizes on the one-graph-per-workgroup kernels, first layer's backward as the dense launch
# (default) vs the per-graph kernel (HCG_NO_XAGG_MID=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
for B in ${@:-40 160 384 768}; do
  for mode in dense pergraph; do
    if [ $mode = pergraph ]; then export HCG_NO_XAGG_MID=1; else unset HCG_NO_XAGG_MID; fi
    ms=$(python $R/bench.py --config REAL --num-graphs $B --steps 300 --warmup 30 --no-cpu-baseline --no-ragged --sustain 0.5 --distinct-batches 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['config'].get('nodes'), d['parity_gate']['worst'])")
    echo "B = $B  $mode: $ms"
  done
done
