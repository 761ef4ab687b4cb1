#!/bin/bash
# pipelined-mode sweep on the GPU box: resolve block size x hot-kernel LDS pad (caps its workgroups per CU)
for lib in "" rb256; do
  [ -n "$lib" ] && export URHGPU_LIB=$PWD/urh_amd/liburhgpu_$lib.so || unset URHGPU_LIB
  timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("lib", "$lib" or "default", "plain     ", "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "latency", d["config"]["single_step_latency_ms"], "value", d["value"])
PY
  for kb in 0 21 28; do
    URH_HOT_LDS_KB=$kb timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --pipeline 2>/dev/null | tail -1 > /tmp/b.json
    python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("lib", "$lib" or "default", "piped lds", $kb, "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "latency", d["config"]["single_step_latency_ms"], "value", d["value"])
PY
  done
done
