#!/bin/bash
# tools/sps10_alone.sh: the tail's kernels over the 10-samples-per-symbol capture (6.7 M rows) on an IDLE machine -- one un-pipelined pass at a
# time under rocprofv3 --kernel-trace --stats: what each kernel takes when no hot kernel runs beside it.  Output: gpurun_out/sps10_alone.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sps10_alone; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/sps10_alone.py <<PY
import sys, torch
sys.path.insert(0, "$R")
import numpy as np
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev, first_segment=0, sps=10)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 1, 10, 0.1, 8, False)
pipe = DevicePipeline(0)
for _ in range(12):
    r = pipe.iq_to_bits(iq, p, want_qad=True)
    torch.cuda.synchronize()
print("rows", len(r.ppseq()))
PY
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o a -- python /tmp/sps10_alone.py > $OUT/log.txt 2>&1)
python3 - $OUT > $R/gpurun_out/sps10_alone.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
PY
cat $R/gpurun_out/sps10_alone.txt; tail -3 $OUT/log.txt
find $OUT -name "*.csv" -size +1M -delete
