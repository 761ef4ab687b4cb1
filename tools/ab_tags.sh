#!/bin/bash
# same-box A/B of tagged builds (python -m urh_amd.build --tag NAME ...): tools/ab_tags.sh NAME [NAME ...]   ("" = the default library)
F="--no-cpu-baseline --no-extra --no-d2h --steps 40 --warmup 5"
for rep in 1 2; do
for t in "default" "$@"; do
  if [ "$t" = "default" ]; then L="X=1"; else L="URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_$t.so"; fi
  env $L python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_unshared'], d['config']['unpipelined_ms_per_step'], d['parity'] if 'parity' in d else '')"
done; done
