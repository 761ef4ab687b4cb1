#!/bin/bash
# rocprofv3 kernel trace of tools/seg_probe.py for a few segmentations -> gpurun_out/seg_<S>_<shape>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "$@"; do
  set -- $cfg; S=$1; SH=${2:-0}
  OUT=$R/gpurun_out/seg_${S}_${SH}; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o b -- python $R/tools/seg_probe.py $S $SH > $OUT/log.txt 2>&1)
  tail -1 $OUT/log.txt > $R/gpurun_out/seg_${S}_${SH}.txt
  python $R/tools/seg_timeline.py $OUT/trace >> $R/gpurun_out/seg_${S}_${SH}.txt 2>&1
  rm -rf $OUT/trace
  cat $R/gpurun_out/seg_${S}_${SH}.txt
done
