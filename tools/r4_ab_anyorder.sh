#!/bin/bash
# round 4: consecutive hot kernels without the AQL barrier bit (hipExtAnyOrderLaunch, tuning "hot_any_order"): headline / device-only step, parity
F="--no-cpu-baseline --no-extra --no-upload --no-pmc --steps 20 --warmup 5"
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
print(sys.argv[1], "headline", d["ms_per_step"], "with pos", c.get("ms_per_step_with_device_positions"), "device only", c.get("device_only_ms_per_step"), "kernel", r["kernel_ms"], "unpipelined", c["unpipelined_ms_per_step"], "single", c.get("single_capture_incl_compact_d2h_ms"), "parity", c.get("parity_bit_exact"))'
for rep in 1 2 3; do
  python bench.py $F 2>/dev/null | python -c "$show" default
  URH_HOT_ANY_ORDER=1 python bench.py $F 2>/dev/null | python -c "$show" any_order
done
URH_HOT_ANY_ORDER=1 python bench.py --no-extra --no-upload --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "$show" any_order_with_parity
OUT=$(pwd)/gpurun_out/r4anyorder; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp URH_HOT_ANY_ORDER=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $OLDPWD/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra --no-upload --no-pmc --no-d2h --no-reference-loop > $OUT/log.txt 2>&1)
python tools/timeline.py $OUT --passes 8 2>&1 | tail -10
find $OUT -name "*kernel_trace.csv" -delete
