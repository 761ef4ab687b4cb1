#!/bin/bash
# round 4: k_me_hist with 2 (default) / 1 / 0 ballot rounds per row before the LDS atomics: estimate and detect_center wall times
for rep in 1 2; do
for t in default h1 h0; do
  if [ "$t" = "default" ]; then L="X=1"; else L="URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_$t.so"; fi
  env $L timeout 200 python tools/est_probe.py 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$t', {k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('estimate_ms','detect_center_ms','center')}, d.get('estimate_stages_ms',{}).get('centers_ms'))"
done; done
