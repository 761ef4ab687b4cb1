#!/bin/bash
TAG=${1:-r3e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for part in ook psk; do
  other=$([ $part = ook ] && echo --no-psk || echo --no-ook)
  (cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$part -o e -- python $R/tools/est_probe.py $other > $OUT/log_$part.txt 2>&1)
  tail -1 $OUT/log_$part.txt | cut -c1-900
  f=$(find $OUT/trace_$part -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && awk -F, 'NR>1 {n=$1; gsub(/"/,"",n); gsub(/urh::/,"",n); gsub(/\(anonymous namespace\)::/,"",n); printf "%-70.70s calls %6s total_us %10.0f avg_us %9.1f\n", n, $2, $3/1000, $4/1000}' $f | sort -k5 -n -r | head -40 > $OUT/kernels_$part.txt
  cat $OUT/kernels_$part.txt
  find $OUT/trace_$part -name "*_trace.csv" -size +3M -delete
done
