#!/bin/bash
# pipelined step time under the inter-hot-kernel knobs (capi.hip): stop event instead of a recorded one, no arena wait (timing only)
mkdir -p gpurun_out
F="--no-cpu-baseline --no-extra --no-d2h --no-reference-loop --steps 40 --warmup 5"
for rep in 1 2; do
for v in "" "URH_HOT_STOP_EVENT=1" "URH_EXP_NO_ARENA_WAIT=1" "URH_HOT_STOP_EVENT=1 URH_EXP_NO_ARENA_WAIT=1" "URH_BENCH_NO_PROFILE=1" "URH_BENCH_NO_PROFILE=1 URH_HOT_STOP_EVENT=1"; do
  echo "== $v" >> gpurun_out/gap_sweep.txt
  env $v python bench.py $F 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline'].get('kernel_ms'), d['value'])" >> gpurun_out/gap_sweep.txt
done
done
cat gpurun_out/gap_sweep.txt
