#!/bin/bash
# Dev tool: run bench.py's N > 1 control flow (captures, exchange hook, barriers, MAX over ranks, rank-0 JSON) with
# 2 ranks on the ONE GPU of a gpurun box: gloo backend, both ranks on device 0.  Timings are meaningless.
set -e
cd "$(dirname "$0")/.."
HCG_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 20 --warmup 5 "$@"
