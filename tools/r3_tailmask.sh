#!/bin/bash
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --gpus 1 --warmup 5 --no-extra --no-cpu-baseline --no-reference-loop "$@" 2>/dev/null | grep '^{' | tail -1 > /tmp/b.json
  python - <<PY
import json
try:
    d=json.loads(open("/tmp/b.json").read())
    print("$label", "stream", d["ms_per_step"], "k", d["roofline"]["kernel_ms"], "dev", d["config"].get("device_only_ms_per_step"), "single", d["config"].get("single_capture_incl_compact_d2h_ms"), d["config"].get("parity_bit_exact"))
except Exception as e:
    print("$label", "failed", e)
PY
}
for rep in 1 2; do
run default A=1 -- --steps 20
run tm4 URH_TAIL_MASKED=1 URH_HOT_CUS_REMOVED=4 -- --steps 20
run tm6 URH_TAIL_MASKED=1 URH_HOT_CUS_REMOVED=6 -- --steps 20
run tm8 URH_TAIL_MASKED=1 URH_HOT_CUS_REMOVED=8 -- --steps 20
run tm12 URH_TAIL_MASKED=1 URH_HOT_CUS_REMOVED=12 -- --steps 20
run r8 URH_HOT_CUS_REMOVED=8 -- --steps 20
done
