#!/bin/bash
# sharded-path validation + per-rank step time of the sharded code path on one GPU (1-rank RCCL group)
out=gpurun_out/${1:-r3s}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "shard" > $out/tests_shard.txt 2>&1; echo "shard tests rc=$?" 
tail -3 $out/tests_shard.txt
for v in default "URH_HOT_CUS_REMOVED=0" ; do
  for rep in 1 2; do
    env $( [ "$v" != default ] && echo $v ) URH_BENCH_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $out/bench_sharded_${v//[^A-Za-z0-9]/_}_$rep.json 2> $out/bench_sharded_err.txt
    python - <<PY
import json
try:
    d=json.loads(open("$out/bench_sharded_${v//[^A-Za-z0-9]/_}_$rep.json").read().strip().splitlines()[-1])
    print("$v", d["ms_per_step"], d["roofline"]["frac"], d["config"].get("parity_bit_exact"))
except Exception as e:
    print("$v", "failed", e)
PY
  done
done
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; echo "all tests rc=$?"
tail -3 $out/tests.txt
