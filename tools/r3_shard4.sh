#!/bin/bash
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29552
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --gpus 1 --warmup 5 --no-extra --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | tail -1 > /tmp/b.json
  python - <<PY
import json
try:
    d=json.loads(open("/tmp/b.json").read())
    print("$label", d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"].get("collectives"), d["config"].get("device_only_ms_per_step"), d["roofline"].get("end_to_end_frac"))
except Exception as e:
    print("$label", "failed", e)
PY
}
for i in 1 2 3; do run shard_k20 URH_BENCH_FORCE_SHARDED=1 -- --steps 20; done
run shard_k20_torch URH_BENCH_FORCE_SHARDED=1 URH_BENCH_TORCH_COLLECTIVES=1 -- --steps 20
run shard_k20_xchg URH_BENCH_FORCE_SHARDED=1 URH_BENCH_HALO_EXCHANGE=1 -- --steps 20
for i in 1 2 3; do run single_k20 A=1 -- --steps 20; done
