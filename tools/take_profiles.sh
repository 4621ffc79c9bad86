#!/bin/bash
# On the MI355X box: every artefact profiles/README.md lists for one tag, into gpurun_out/<tag>_* (copy into profiles/ afterwards).
# usage: tools/take_profiles.sh r02_a [quick]
set -o pipefail
tag=${1:?tag}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.log || exit 1
echo "bench done"
PROF="--steps 100 --warmup 20 --no-cpu-baseline --no-ragged --sustain 0.3 --distinct-batches 4 --no-parity-gate"
for cfg in C3 C5 REAL RAGGED; do
  rm -rf /tmp/p_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python $R/bench.py --config $cfg $PROF > /dev/null 2> $O/${tag}_stats_$cfg.log || exit 1
  cp "$(find /tmp/p_stats -name '*kernel_stats.csv' | head -1)" $O/${tag}_kernel_stats_$cfg.csv
  echo "kernel stats $cfg done"
done
cp $O/${tag}_kernel_stats_C3.csv $O/${tag}_kernel_stats.csv
PMC="--steps 10 --warmup 5 --no-cpu-baseline --no-ragged --no-graph --sustain 0 --distinct-batches 2 --no-parity-gate"
rm -f $O/${tag}_traffic.json
for cfg in C3 C5 REAL; do
  rm -rf /tmp/p_fetch /tmp/p_write
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python $R/bench.py --config $cfg $PMC > /dev/null 2> $O/${tag}_fetch_$cfg.log || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python $R/bench.py --config $cfg $PMC > /dev/null 2> $O/${tag}_write_$cfg.log || exit 1
  python $R/tools/pmc_traffic.py /tmp/p_fetch /tmp/p_write $O/${tag}_traffic.json $cfg > $O/${tag}_traffic_$cfg.txt
  echo "pmc $cfg done"
done
# SQ counters of the C3 kernels (the headline's limiter measured, not argued): wave cycles, waits, instruction mix
rm -rf /tmp/p_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/p_sq -- python $R/bench.py --config C3 $PMC > /dev/null 2> $O/${tag}_sq_C3.log || exit 1
python $R/tools/pmc_sq.py /tmp/p_sq > $O/${tag}_sq_c3.txt
rm -rf /tmp/p_sq2
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/p_sq2 -- python $R/bench.py --config C3 $PMC > /dev/null 2> $O/${tag}_sq2_C3.log && python $R/tools/pmc_sq.py /tmp/p_sq2 >> $O/${tag}_sq_c3.txt
echo "sq C3 done"
[ "$2" = quick ] && { echo "all done (quick)"; exit 0; }
python $R/bench.py --forward-only --no-cpu-baseline > $O/${tag}_bench_forward_only_C2.json 2>/dev/null || exit 1
python $R/bench.py --config REAL --steps 100 --warmup 10 --no-cpu-baseline > $O/${tag}_bench_REAL.json 2>/dev/null || exit 1
python $R/bench.py --config REAL40 --steps 200 --warmup 20 --no-cpu-baseline --sustain 2 > $O/${tag}_bench_REAL40.json 2>/dev/null || exit 1
python $R/bench.py --config C5 --steps 50 --warmup 5 --no-cpu-baseline > $O/${tag}_bench_C5.json 2>/dev/null || exit 1
python $R/bench.py --config RAGGED --steps 100 --warmup 10 --no-cpu-baseline > $O/${tag}_bench_RAGGED.json 2>/dev/null || exit 1
echo "all done"
