F="--no-cpu-baseline --no-extra --no-upload --no-pmc --steps 20 --warmup 5"
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
print(sys.argv[1], "headline", d["ms_per_step"], "with pos", c.get("ms_per_step_with_device_positions"), "device only", c.get("device_only_ms_per_step"), "kernel", r["kernel_ms"])'
for rep in 1 2; do
for r in 4 5 6 8; do
  URH_HOT_CUS_REMOVED=$r python bench.py $F 2>/dev/null | python -c "$show" removed_$r
done; done
