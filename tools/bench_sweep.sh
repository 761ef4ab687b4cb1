#!/bin/bash
# tools/bench_sweep.sh <tail_cus...>: bench.py with different CU partitions (GPU box)
for t in "$@"; do
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --tail-cus $t 2>&1 | tail -1 > /tmp/b.json
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("tail_cus", $t, "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "latency", d["config"]["single_step_latency_ms"], "value", d["value"])
PY
done
