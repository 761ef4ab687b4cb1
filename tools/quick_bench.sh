#!/bin/bash
# pipelined step time of the single-GPU path and of the sharded path over a 1-rank RCCL group (GPU box), two runs each
mkdir -p gpurun_out
F="--no-cpu-baseline --no-extra --no-d2h --steps 40 --warmup 5"
show='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d["roofline"].get("kernel_ms"), d["config"].get("unpipelined_ms_per_step"), d["value"])'
for rep in 1 2; do
  echo "== single" >> gpurun_out/quick_bench.txt
  python bench.py $F 2>/dev/null | python -c "$show" >> gpurun_out/quick_bench.txt
  echo "== sharded (1-rank RCCL)" >> gpurun_out/quick_bench.txt
  URH_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2951$rep python bench.py $F 2>/dev/null | python -c "$show" >> gpurun_out/quick_bench.txt
done
cat gpurun_out/quick_bench.txt
