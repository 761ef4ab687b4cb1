#!/bin/bash
# bench.py's single-GPU pipelined step under URH_HOT_LDS_KB (hot workgroups per CU: 0 -> 7, 21 -> 6, 27 -> 5)
F="--no-cpu-baseline --no-extra --no-d2h --no-reference-loop --steps 40 --warmup 5"
for rep in 1 2; do
for pad in 0 21 27; do
  URH_HOT_LDS_KB=$pad python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('pad $pad', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
