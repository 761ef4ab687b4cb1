#!/bin/bash
# round 4: consecutive hot kernels of direct passes on two alternating masked streams, the next one released by a gate at 97 % (hot_overlap)
timeout 600 python -m pytest tests/test_stream_segments.py -x -q -m gpu -k "direct" 2>&1 | tail -3
F="--no-extra --no-upload --no-pmc --steps 20 --warmup 5"
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
print(sys.argv[1], "headline", d["ms_per_step"], "with pos", c.get("ms_per_step_with_device_positions"), "device only", c.get("device_only_ms_per_step"), "kernel", r["kernel_ms"], "parity", c.get("parity_bit_exact"))'
for rep in 1 2 3; do
  python bench.py $F --no-cpu-baseline 2>/dev/null | python -c "$show" default
  URH_HOT_OVERLAP=1 python bench.py $F --no-cpu-baseline 2>/dev/null | python -c "$show" overlap_97
  URH_HOT_OVERLAP=1 URH_HOT_OVERLAP_PCT=90 python bench.py $F --no-cpu-baseline 2>/dev/null | python -c "$show" overlap_90
done
URH_HOT_OVERLAP=1 python bench.py $F 2>/dev/null | python -c "$show" overlap_97_with_parity
