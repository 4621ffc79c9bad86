#!/bin/bash
# dev tool: un-profiled eager ms/step of the default bench, alternating between environment settings on ONE box
# usage: tools/ab_env.sh ROUNDS "ENV1" "ENV2" ...      (an empty string = the default build)
rounds=$1; shift
for i in $(seq $rounds); do
  for e in "$@"; do
    r=$(env $e python /root/repo/bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 >/dev/null | grep -E "timed eager|hipGraph replay \(full" | sed 's/.*: //' | tr '\n' ' ')
    echo "round $i [${e:-default}] eager / replay: $r"
  done
done
