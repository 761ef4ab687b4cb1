#!/bin/bash
# sharded code path on one GPU (1-rank RCCL group): host-time probe + kernel timeline
TAG=${1:-r3sp}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
MASTER_PORT=29541 timeout 300 python $R/tools/shard_host_probe.py > $OUT/host_probe.json 2> $OUT/host_probe.err; cat $OUT/host_probe.json
(cd /tmp && TMPDIR=/tmp URH_BENCH_FORCE_SHARDED=1 MASTER_PORT=29542 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o b -- python $R/bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-extra > $OUT/log.txt 2>&1)
tail -1 $OUT/log.txt | cut -c1-300
python $R/tools/timeline.py $OUT/trace --passes 8 --all > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
find $OUT/trace -name "*_trace.csv" -size +3M -delete
