#!/bin/bash
# Collect rocprofv3 kernel-trace stats and PMC counters for the hot kernel (run on the GPU box via gpurun).
# usage: tools/prof_pmc.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{stats,pmc1..4}
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-d2h --no-pipeline $*"   # --no-pipeline: no extra pipelined passes in the trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $R/bench.py $ARGS > $OUT/stats.log 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_demod|k_afp|k_fir|k_costas" --output-format csv -d $OUT/pmc$i -o b -- python $R/bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
find $OUT -name "*.csv" | head -20
