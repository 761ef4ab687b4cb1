#!/bin/bash
# PMC counters of the tail kernels (GPU box): tools/prof_pmc_tail.sh <tag> [bench args]; summary with tools/pmc_summary.py
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-d2h --no-pipeline --torch-capture $*"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_emit_rows_tiles|k_expand_tiles|k_resolve_one|k_scan_lookback|k_tile_scan" --output-format csv -d $OUT/pmc$i -o b -- python $R/bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $OUT | grep -v "^==" | awk '{print}' | sed 's/  */ /g' | sort | head -150
