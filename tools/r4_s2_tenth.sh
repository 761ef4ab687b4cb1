#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
OUT=$R/gpurun_out/r4up8; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o b -- python $R/tools/r4_upload_probe.py 8 > $OUT/log.txt 2>&1)
grep "upload in" $OUT/log.txt
python tools/r4_upload_trace.py $OUT/trace > gpurun_out/r4s2_up8_trace.txt 2>&1; rm -rf $OUT/trace
head -120 gpurun_out/r4s2_up8_trace.txt
