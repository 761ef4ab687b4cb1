#!/bin/bash
TAG=${1:-r3e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for part in ook psk; do
  other=$([ $part = ook ] && echo --no-psk || echo --no-ook)
  (cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$part -o e -- python $R/tools/est_probe.py $other > $OUT/log_$part.txt 2>&1)
  tail -1 $OUT/log_$part.txt | cut -c1-900
  f=$(find $OUT/trace_$part -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $OUT/kernels_$part.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:40]:
    n = r["Name"].replace("urh::", "").replace("(anonymous namespace)::", "")
    print(f"{n[:70]:70s} calls {int(r['Calls']):6d} total_us {float(r['TotalDurationNs']) / 1e3:10.0f} avg_us {float(r['AverageNs']) / 1e3:9.1f}")
PY
  cat $OUT/kernels_$part.txt
  find $OUT/trace_$part -name "*_trace.csv" -size +3M -delete
done
