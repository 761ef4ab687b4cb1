#!/bin/bash
# pipelined / un-pipelined step of the default library + the sharded step (1-rank RCCL), twice
F="--no-cpu-baseline --no-extra --no-d2h --steps 40 --warmup 5"
for rep in 1 2; do
  python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single ', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_unshared'], d['config']['unpipelined_ms_per_step'])"
  URH_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2953$rep python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('sharded', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
