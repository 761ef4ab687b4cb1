#!/bin/bash
# rocprofv3 kernel stats of the default bench (GPU box): tools/prof_stats.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-d2h --no-pipeline "$@" > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    tot = 0
    for r in csv.DictReader(open(f)):
        if "urh::" in r["Name"]:
            print(f'{r["Name"][:70]:70s} {int(r["Calls"]):4d} {float(r["AverageNs"])/1e3:9.2f} us')
            tot += float(r["AverageNs"])
    print("sum of averages", tot / 1e3, "us")
PY
tail -1 $OUT/log.txt | cut -c1-400
