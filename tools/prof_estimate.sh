#!/bin/bash
# rocprofv3 kernel stats of the estimate chain on the configs[2] capture (GPU box): tools/prof_estimate.sh <tag>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/est_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $R/tools/prof_estimate.py "$@" > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:40]:
        print(f'{r["Name"][:90]:90s} {int(r["Calls"]):4d} {float(r["AverageNs"])/1e3:9.2f} us')
PY
tail -2 $OUT/log.txt | cut -c1-600
