#!/bin/bash
# same-box A/B of the working tree's library against urh_amd/liburhgpu_prev.so (developer tool, GPU box)
for rep in 1 2 3; do
for t in "" prev; do
  if [ -z "$t" ]; then L="X=1"; else L="URHGPU_LIB=/root/repo/urh_amd/liburhgpu_$t.so"; fi
  env $L python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${t:-new }', d['ms_per_step'], d['roofline']['kernel_ms'], round(d['ms_per_step']-d['roofline']['kernel_ms'],4))"
done; done
for t in "" prev; do
  if [ -z "$t" ]; then L="X=1"; else L="URHGPU_LIB=/root/repo/urh_amd/liburhgpu_$t.so"; fi
  echo "${t:-new}"; env $L python tools/dtype_probe.py 2>/dev/null | cut -c1-40
done
