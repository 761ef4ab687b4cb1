#!/bin/bash
# session 2, second call: timelines of a streamed single capture (rocprofv3 kernel trace) for a few segmentations; the failing fuzz test again
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_signal_shim.py -x -q -m gpu > gpurun_out/r4s2_pytest2.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r4s2_pytest2.txt
SB=3 bash tools/r4_seg_prof.sh "6 0" "8 0" > /dev/null 2>&1
SB=2 bash tools/r4_seg_prof.sh "4 0" > /dev/null 2>&1
for f in gpurun_out/r4seg_6_0.txt gpurun_out/r4seg_8_0.txt gpurun_out/r4seg_4_0.txt; do echo "#### $f"; head -75 $f; done
