#!/bin/bash
# rocprofv3 kernel stats of the sharded bench path with a 1-rank RCCL group (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stats_sharded
mkdir -p $OUT
export TMPDIR=/tmp URH_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 10 and int(r["Calls"]) <= 60]
    tot = 0
    for r in rows:
        per_step = float(r["TotalDurationNs"]) / 17 / 1e3
        print(f'{r["Name"][:80]:80s} {int(r["Calls"]):4d} avg {float(r["AverageNs"])/1e3:8.2f} us  per-step {per_step:8.2f} us')
        tot += per_step
    print("sum per step", tot, "us")
PY
