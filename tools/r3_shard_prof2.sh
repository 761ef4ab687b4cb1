#!/bin/bash
TAG=${1:-r3sp2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 URH_BENCH_FORCE_SHARDED=1 MASTER_PORT=29542
for v in xchg given; do
(cd /tmp && TMPDIR=/tmp URH_BENCH_HALO_EXCHANGE=$([ $v = xchg ] && echo 1 || echo 0) timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o b -- python $R/bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-extra > $OUT/log_$v.txt 2>&1)
python $R/tools/timeline.py $OUT/trace_$v --passes 5 --all > $OUT/timeline_$v.txt 2>&1
cat $OUT/timeline_$v.txt
find $OUT/trace_$v -name "*_trace.csv" -size +3M -delete
done
