#!/bin/bash
# w8 build (64 VGPRs, 8 waves per SIMD possible) with the hot workgroups' LDS padded so that 7 fit per CU: one wave slot and 64 VGPRs per SIMD stay free for the tail
F="--no-cpu-baseline --no-extra --no-d2h --steps 40 --warmup 5"
for rep in 1 2; do
for v in "X=1" "URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_w8.so URH_HOT_LDS_KB=20" "URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_w8.so URH_HOT_LDS_KB=19" "URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_w8.so URH_HOT_LDS_KB=24"; do
  env $v python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v'[-40:], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_unshared'], d['config']['unpipelined_ms_per_step'])"
done; done
