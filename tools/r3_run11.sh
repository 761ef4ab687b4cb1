#!/bin/bash
# sharded per-rank step (1-rank RCCL group on one GPU) with the CU-masked hot stream: LDS padding of the hot workgroups 33 KiB (default so far) / 0 / 16, mask 4 / 0
mkdir -p gpurun_out/r3n
F="--no-cpu-baseline --no-extra --no-d2h --steps 40 --warmup 5"
for rep in 1 2; do
for v in "X=1" "URH_HOT_LDS_KB=0" "URH_HOT_LDS_KB=16" "URH_HOT_CUS_REMOVED=0" "URH_HOT_CUS_REMOVED=0 URH_HOT_LDS_KB=0" "URH_HOT_CUS_REMOVED=6 URH_HOT_LDS_KB=0"; do
  env $v URH_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2954$rep python bench.py $F 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('sharded %-44s' % '$v', d['ms_per_step'], 'k', d['roofline']['kernel_ms'])" | tee -a gpurun_out/r3n/sharded.txt
done; done
