#!/bin/bash
# round 4: DIRECT passes (stream_policy 3 / 4: one segment, the tail's kernels store rows and packed results into the pinned host blob) against
# the pack + copy-engine route (policy 0 for back-to-back passes)
timeout 600 python -m pytest tests/test_stream_segments.py -x -q -m gpu -k "direct or chain or interleave" 2>&1 | tail -3
F="--no-cpu-baseline --no-extra --no-upload --no-pmc --steps 20 --warmup 5"
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
print(sys.argv[1], "headline", d["ms_per_step"], "with pos", c.get("ms_per_step_with_device_positions"), "device only", c.get("device_only_ms_per_step"), "kernel", r["kernel_ms"], "single", c.get("single_capture_incl_compact_d2h_ms"), c.get("stream_stats"))'
for rep in 1 2 3; do
  URH_STREAM_POLICY=0 python bench.py $F 2>/dev/null | python -c "$show" policy0
  URH_STREAM_POLICY=5 python bench.py $F 2>/dev/null | python -c "$show" policy5_default
  URH_STREAM_POS_DIRECT=0 python bench.py $F 2>/dev/null | python -c "$show" policy5_pos_via_pack
done
