#!/bin/bash
# arena reuse guarded on the host (default) against a stream-level wait (URH_ARENA_WAIT=stream), single-GPU and sharded pipelined steps
F="--no-cpu-baseline --no-extra --no-d2h --steps 40 --warmup 5"
for rep in 1 2 3; do
for v in "X=1" "URH_ARENA_WAIT=stream"; do
  env $v python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single  $v', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['unpipelined_ms_per_step'])"
done; done
for v in "X=1" "URH_ARENA_WAIT=stream"; do
  env $v URH_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('sharded $v', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
