#!/bin/bash
# dev tool (GPU box): SQ counters of one bench config's kernels, two rocprofv3 --pmc passes -> stdout (tools/pmc_sq.py).
# usage: tools/sq_config.sh REAL
R=${GRAFT_REPO_ROOT:-/root/repo}
cfg=${1:?config}
cd /tmp && export TMPDIR=/tmp
PMC="--steps 10 --warmup 5 --no-cpu-baseline --no-ragged --no-graph --sustain 0 --distinct-batches 2 --no-parity-gate"
rm -rf /tmp/p_sq /tmp/p_sq2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/p_sq -- python $R/bench.py --config $cfg $PMC > /dev/null 2> /tmp/sq1.log || { tail -5 /tmp/sq1.log; exit 1; }
python $R/tools/pmc_sq.py /tmp/p_sq
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/p_sq2 -- python $R/bench.py --config $cfg $PMC > /dev/null 2> /tmp/sq2.log || { tail -5 /tmp/sq2.log; exit 1; }
python $R/tools/pmc_sq.py /tmp/p_sq2
