#!/bin/bash
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 URH_BENCH_FORCE_SHARDED=1 MASTER_PORT=29552
run() { # label, env..., -- args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --gpus 1 --warmup 5 --no-extra --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | tail -1 > /tmp/b.json
  python - <<PY
import json
try:
    d=json.loads(open("/tmp/b.json").read())
    print("$label", d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["collectives"], d["config"]["single_step_latency_ms"])
except Exception as e:
    print("$label", "failed", e)
PY
}
run k20 A=1 -- --steps 20
run k20_noprof URH_BENCH_NO_PROFILE=1 -- --steps 20
run k100 A=1 -- --steps 100
run k400 A=1 -- --steps 400
run k400_noprof URH_BENCH_NO_PROFILE=1 -- --steps 400
run k20_torch URH_BENCH_TORCH_COLLECTIVES=1 -- --steps 20
run k400_torch URH_BENCH_TORCH_COLLECTIVES=1 -- --steps 400
run k20 A=1 -- --steps 20
