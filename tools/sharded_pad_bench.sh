#!/bin/bash
# bench.py's sharded step (1-rank RCCL) under URH_HOT_LDS_KB
F="--no-cpu-baseline --no-extra --no-d2h --steps 40 --warmup 5"
p=29800
for rep in 1 2; do
for pad in 0 21 27 33; do
  p=$((p+1))
  URH_HOT_LDS_KB=$pad URH_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$p python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('pad $pad', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
