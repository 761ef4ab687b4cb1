#!/bin/bash
# final pair for profiles/: the default bench line and the rocprofv3 kernel stats of the same (pipelined) command, same box: tools/prof_final.sh <tag>
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
OUTP=$R/gpurun_out/stats_${TAG}_pipelined; mkdir -p $OUTP
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUTP -o b -- python $R/bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-extra --no-d2h --no-reference-loop > $OUTP/log.txt 2>&1)
OUTU=$R/gpurun_out/stats_${TAG}_unpipelined; mkdir -p $OUTU
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUTU -o b -- python $R/bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-extra --no-d2h --no-pipeline > $OUTU/log.txt 2>&1)
find $OUTP $OUTU -name "*kernel_trace.csv" -delete
ls $OUTP $OUTU
