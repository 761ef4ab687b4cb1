#!/bin/bash
# sharded pipelined step (1-rank RCCL) under the context knobs: LDS padding of the hot workgroups, tail stream priority
mkdir -p gpurun_out
F="--no-cpu-baseline --no-extra --no-d2h --steps 40 --warmup 5"
show='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d["roofline"].get("kernel_ms"), d["value"])'
i=0
for v in "URH_X=0" "URH_HOT_LDS_KB=21" "URH_HOT_LDS_KB=33" "URH_TAIL_PRIORITY=1" "URH_TAIL_PRIORITY=1 URH_HOT_LDS_KB=21" "URH_X=0"; do
  i=$((i+1))
  echo "== $v" >> gpurun_out/sharded_knobs.txt
  env $v URH_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2952$i python bench.py $F 2>/dev/null | python -c "$show" >> gpurun_out/sharded_knobs.txt
done
cat gpurun_out/sharded_knobs.txt
