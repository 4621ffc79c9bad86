#!/bin/bash
# dev tool: interleaved A/B of prebuilt library variants (hcatgnet_amd/csrc/_variants/<name>.so, loaded through HCG_LIB) on ONE box:
# un-profiled bench lines (sustained rotation, hipGraph) per config.  usage: tools/ab_lib.sh "CFG1 CFG2" A B A B
R=${GRAFT_REPO_ROOT:-/root/repo}
cfgs=$1; shift
for v in "$@"; do
  for c in $cfgs; do
    HCG_LIB=$R/hcatgnet_amd/csrc/_variants/$v.so python $R/bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-ragged --sustain 1.5 \
      > /tmp/ab_${v}_$c.json 2> /tmp/ab_${v}_$c.log || { echo "$v $c FAILED"; tail -5 /tmp/ab_${v}_$c.log; exit 1; }
    python - /tmp/ab_${v}_$c.json "$v" "$c" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:10s} {sys.argv[3]:8s} {d['ms_per_step']:.4f} ms/step  sustained {d['sustained']['ms_per_step']:.4f}  burst graph {d['burst']['hipgraph_ms_per_step']}  eager {d['burst']['eager_ms_per_step']:.4f}")
PY
  done
done
