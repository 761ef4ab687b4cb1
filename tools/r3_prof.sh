#!/bin/bash
# kernel (+ memory copy) trace of the stream-mode bench; summary + timeline into gpurun_out/$1
TAG=${1:-r3p}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace ${PROF_COPY:+--memory-copy-trace} --stats --output-format csv -d $OUT/trace -o b -- python $R/bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-extra --no-reference-loop "$@" > $OUT/log.txt 2>&1)
tail -2 $OUT/log.txt | cut -c1-600
python $R/tools/timeline.py $OUT/trace --passes 14 > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep "urh::" $f | awk -F, '{n=$1; gsub(/urh::/,"",n); printf "%-60.60s calls %6s avg %10.0f ns\n", n, $2, $4}' | head -20 > $OUT/kernel_stats.txt
cat $OUT/kernel_stats.txt
find $OUT/trace -name "*_trace.csv" -size +3M -delete
