#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r4_second.txt
: > $O
timeout 900 python -m pytest tests/test_stream_segments.py tests/test_capture_stream.py -x -q -m gpu > gpurun_out/r4_second_pytest.txt 2>&1
echo "pytest rc=$?" >> $O
tail -15 gpurun_out/r4_second_pytest.txt >> $O
for cfg in "8 0 3" "8 0 2" "6 0 3" "5 1 3" "4 0 2" "1 0 1"; do
  set -- $cfg
  SB=$3 timeout 200 python tools/r4_seg_probe.py $1 $2 8 40 2>&1 | tail -1 >> $O
done
cat $O
SB=3 bash tools/r4_seg_prof.sh "8 0"
