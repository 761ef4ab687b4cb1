#!/bin/bash
# the sharded code path on a 1-rank RCCL group (what every rank of `bench.py --gpus N` runs): step with the rank's compact blob copied
# to the host (the headline of N > 1 runs), device-only step, hot kernel -- with RCCL called directly / through torch.distributed, halo
# handed over with the shard / exchanged.  -> gpurun_out/<tag>/sharded.txt
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
: > gpurun_out/$TAG/sharded.txt
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
for v in halo_with_shard_rccl:A=1 halo_with_shard_torch_collectives:URH_BENCH_TORCH_COLLECTIVES=1 halo_exchanged_torch_collectives:URH_BENCH_HALO_EXCHANGE=1,URH_BENCH_TORCH_COLLECTIVES=1; do
  label=${v%%:*}; envs=${v#*:}
  for rep in 1 2; do
    env ${envs//,/ } URH_BENCH_FORCE_SHARDED=1 MASTER_PORT=29571 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > /tmp/sb.json
    python - >> gpurun_out/$TAG/sharded.txt <<PY
import json
try:
    d = json.loads(open("/tmp/sb.json").read())
    c = d["config"]
    print("$label", "ms_per_step_with_d2h", d["ms_per_step"], "device_only", c["device_only_ms_per_step"], "hot_kernel_ms", d["roofline"]["kernel_ms"],
          "frac", d["roofline"]["frac"], c["collectives"], "all_gathers_per_pass", c["all_gathers_per_pass"], "d2h_bytes", c.get("d2h_bytes_per_step"),
          "host_blob_equals_device", c.get("host_blob_equals_device_outputs"))
except Exception as e:
    print("$label", "failed", e)
PY
  done
done
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR
cat gpurun_out/$TAG/sharded.txt
