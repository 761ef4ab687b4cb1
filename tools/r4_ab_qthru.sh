#!/bin/bash
# round 4: qad stores written through (agent-scope) instead of non-temporal: kernel time and the gap between two hot kernels
F="--no-cpu-baseline --no-extra --no-upload --no-pmc --steps 20 --warmup 5"
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
print(sys.argv[1], "headline", d["ms_per_step"], "device only", c.get("device_only_ms_per_step"), "kernel", r["kernel_ms"], "unshared", r["kernel_ms_unshared"], "unpipelined", c["unpipelined_ms_per_step"])'
for rep in 1 2; do
  python bench.py $F 2>/dev/null | python -c "$show" default
  URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_qthru.so python bench.py $F 2>/dev/null | python -c "$show" qad_through
done
OUT=$(pwd)/gpurun_out/r4qthru; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp URHGPU_LIB=$OLDPWD/urh_amd/liburhgpu_qthru.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $OLDPWD/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra --no-upload --no-pmc --no-d2h --no-reference-loop > $OUT/log.txt 2>&1)
python tools/timeline.py $OUT --passes 6 2>&1 | tail -8
find $OUT -name "*kernel_trace.csv" -delete
