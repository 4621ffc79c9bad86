#!/bin/bash
# On the MI355X box: every artefact profiles/README.md lists for one tag, into gpurun_out/<tag>_* (copy into profiles/ afterwards).
# usage: tools/take_profiles.sh r01_h
set -o pipefail
tag=${1:?tag}
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.log || exit 1
echo "bench done"
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2> $O/${tag}_stats.log || exit 1
cp "$(find /tmp/p_stats -name '*kernel_stats.csv' | head -1)" $O/${tag}_kernel_stats.csv
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-graph > /dev/null 2> $O/${tag}_fetch.log || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-graph > /dev/null 2> $O/${tag}_write.log || exit 1
python $R/tools/pmc_traffic.py /tmp/p_fetch /tmp/p_write $O/${tag}_traffic.json
echo "pmc done"
python $R/bench.py --forward-only --no-cpu-baseline > $O/${tag}_bench_forward_only_C2.json 2>/dev/null || exit 1
python $R/bench.py --config REAL --steps 100 --warmup 10 --no-cpu-baseline > $O/${tag}_bench_REAL.json 2>/dev/null || exit 1
python $R/bench.py --config REAL40 --steps 200 --warmup 20 --no-cpu-baseline > $O/${tag}_bench_REAL40.json 2>/dev/null || exit 1
python $R/bench.py --config C5 --steps 50 --warmup 5 --no-cpu-baseline --roofline-entry hcg_mid_layer_bwd > $O/${tag}_bench_C5.json 2>/dev/null || exit 1
echo "all done"
