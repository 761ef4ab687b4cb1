#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "shard or rccl" 2>&1 | tail -6
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29552
for i in 1 2 3; do
URH_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --gpus 1 --warmup 5 --steps 20 --no-extra --no-cpu-baseline 2>/tmp/err.txt | grep '^{' | tail -1 > /tmp/b.json
python - <<PY
import json
try:
    d=json.loads(open("/tmp/b.json").read())
    c=d["config"]
    print("sharded", d["metric"][-45:], d["ms_per_step"], "dev", c["device_only_ms_per_step"], "k", d["roofline"]["kernel_ms"], c.get("host_blob_equals_device_outputs"), c.get("d2h_bytes_per_step"), d["roofline"].get("end_to_end_frac"))
except Exception as e:
    print("failed", e); print(open("/tmp/err.txt").read()[-1500:])
PY
done
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('single', d['ms_per_step'], d['config']['device_only_ms_per_step'], d['config']['parity_bit_exact'])"
