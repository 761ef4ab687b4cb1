#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r4s2_upload2.txt; : > $O
timeout 300 python tools/r4_upload_probe.py 8 4 16 2 8 16 2>&1 | grep -v amdgpu.ids | grep -v "resident single" >> $O
echo "--- fresh processes" >> $O
for pc in 8 16 4; do timeout 100 python tools/r4_upload_probe.py $pc 2>&1 | grep "upload in" >> $O; done
cat $O
timeout 600 python -m pytest tests/test_stream_segments.py -x -q -m gpu -k upload > gpurun_out/r4s2_pytest9.txt 2>&1
echo "pytest upload rc=$?"; tail -3 gpurun_out/r4s2_pytest9.txt
timeout 600 python -m pytest tests/test_full_size.py -x -q -m gpu -k "streamed" > gpurun_out/r4s2_pytest9b.txt 2>&1
echo "pytest full size rc=$?"; tail -3 gpurun_out/r4s2_pytest9b.txt
