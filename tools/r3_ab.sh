#!/bin/bash
# same-box A/B of tagged builds / env knobs in stream mode: tools/r3_ab.sh <outfile> <reps> "<tag>[:ENV=val[,ENV=val]]" ...
OUT=$1; REPS=$2; shift 2
for rep in $(seq $REPS); do
for spec in "$@"; do
  t=${spec%%:*}; envs=""; [ "$spec" != "$t" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  if [ "$t" = "default" ]; then L="X=1"; else L="URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_$t.so"; fi
  env $L $envs python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('%-28s' % '$spec', 'stream', d['ms_per_step'], 'withpos', c.get('ms_per_step_with_device_positions'), 'dev', c['device_only_ms_per_step'], 'k', r['kernel_ms'], 'k_alone', r['kernel_ms_unshared'], 'unpiped', c['unpipelined_ms_per_step'], 'single', c.get('single_capture_incl_compact_d2h_ms'), 'lat', c['single_step_latency_ms'], c.get('host_loop'))" | tee -a $OUT
done; done
