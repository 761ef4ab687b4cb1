#!/bin/bash
# round-3 record for profiles/: the default bench line, rocprofv3 kernel stats of the device-only pipelined loop and of passes one after
# the other, the timeline of the pipelined passes, PMC counters (tools/prof_pmc.sh): tools/r3_final_profiles.sh <tag>
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$TAG
python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -c 600 gpurun_out/$TAG/bench.err
for mode in pipelined unpipelined; do
  OUT=$R/gpurun_out/$TAG/stats_$mode; mkdir -p $OUT
  if [ $mode = pipelined ]; then A="--no-d2h --no-reference-loop"; else A="--no-d2h --no-pipeline"; fi
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra $A > $OUT/log.txt 2>&1)
  python tools/timeline.py $OUT --passes 12 > gpurun_out/$TAG/timeline_$mode.txt 2>&1
  find $OUT -name "*kernel_trace.csv" -delete
done
bash tools/prof_pmc.sh $TAG > gpurun_out/$TAG/pmc_files.txt 2>&1
python tools/prof_collect.py gpurun_out/prof_$TAG gpurun_out/$TAG/fsk_1gib > gpurun_out/$TAG/pmc_summary.txt 2>&1
find gpurun_out/prof_$TAG -name "*.csv" -size +2M -delete
tail -5 gpurun_out/$TAG/pmc_summary.txt
cat gpurun_out/$TAG/timeline_pipelined.txt | tail -4
