#!/bin/bash
# round-3 GPU check 2: timeline of the stream mode (default build), A/B with the 8-waves-per-SIMD build
mkdir -p gpurun_out/r3b
bash tools/r3_prof.sh r3b/prof_default
echo "=== A/B default vs w8 (stream mode: value with D2H; device-only; kernel in-run; kernel unshared; unpipelined)"
for rep in 1 2; do
for t in default w8; do
  if [ "$t" = "default" ]; then L="X=1"; else L="URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_$t.so"; fi
  env $L python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('$t', 'stream', d['ms_per_step'], 'nopos', c.get('ms_per_step_without_positions'), 'dev', c['device_only_ms_per_step'], 'k', r['kernel_ms'], 'k_alone', r['kernel_ms_unshared'], 'unpiped', c['unpipelined_ms_per_step'], 'single', c.get('single_capture_incl_compact_d2h_ms'))" | tee -a gpurun_out/r3b/ab.txt
done; done
URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_w8.so bash tools/r3_prof.sh r3b/prof_w8
