#!/bin/bash
# round 4: the 8-waves-per-SIMD build of the hot kernel (64 VGPRs) WITH the CU-masked hot stream (round 3), which it was never combined with:
# the headline loop (stream, D2H) and the device-only loop
F="--no-cpu-baseline --no-extra --no-upload --no-pmc --steps 40 --warmup 5"
for rep in 1 2 3; do
for t in default w8; do
  if [ "$t" = "default" ]; then L="X=1"; else L="URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_$t.so"; fi
  env $L python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('$t', 'headline', d['ms_per_step'], 'with pos', c.get('ms_per_step_with_device_positions'), 'device only', c.get('device_only_ms_per_step'), 'kernel', r['kernel_ms'], 'unshared', r['kernel_ms_unshared'], 'unpipelined', c['unpipelined_ms_per_step'], 'single', c.get('single_capture_incl_compact_d2h_ms'))"
done; done
