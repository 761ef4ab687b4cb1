#!/bin/bash
out=gpurun_out/${1:-r3s3}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "shard or rccl" > $out/tests_shard.txt 2>&1; echo "shard tests rc=$?"
tail -3 $out/tests_shard.txt
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
for cfg in "COMM=rccl HALO=given" "COMM=rccl HALO=exchange" "COMM=torch HALO=given"; do
  env $cfg MASTER_PORT=29551 timeout 300 python tools/shard_host_probe.py 2>/dev/null | tail -1
done
for v in "A=1" "URH_BENCH_TORCH_COLLECTIVES=1" "URH_BENCH_HALO_EXCHANGE=1"; do
  for rep in 1 2; do
    env $v URH_BENCH_FORCE_SHARDED=1 MASTER_PORT=29552 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>$out/bench_err.txt | tail -1 > $out/b.json
    python - <<PY
import json
try:
    d=json.loads(open("$out/b.json").read())
    print("$v", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["config"]["collectives"], d["config"]["all_gathers_per_pass"])
except Exception as e:
    print("$v", "failed", e)
PY
  done
done
