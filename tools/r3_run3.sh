#!/bin/bash
# round-3 GPU check 3: tail with 16 chunks per wavefront -- parity, A/B of chunks per wavefront and 8 waves per SIMD, timeline
mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_parity.py tests/test_capture_stream.py -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/r3c/tests.txt
echo "=== A/B (stream mode: value with D2H; without positions; device-only; kernel in-run; kernel unshared; unpipelined; single capture)"
for rep in 1 2; do
for t in default cpw8 cpw32 w8; do
  if [ "$t" = "default" ]; then L="X=1"; else L="URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_$t.so"; fi
  env $L python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('$t', 'stream', d['ms_per_step'], 'nopos', c.get('ms_per_step_without_positions'), 'dev', c['device_only_ms_per_step'], 'k', r['kernel_ms'], 'k_alone', r['kernel_ms_unshared'], 'unpiped', c['unpipelined_ms_per_step'], 'single', c.get('single_capture_incl_compact_d2h_ms'), 'lat', c['single_step_latency_ms'])" | tee -a gpurun_out/r3c/ab.txt
done; done
bash tools/r3_prof.sh r3c/prof_default
URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_w8.so bash tools/r3_prof.sh r3c/prof_w8
