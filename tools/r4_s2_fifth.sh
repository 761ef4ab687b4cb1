#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/r4_upload_dbg.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4s2_upload_dbg.txt; cat gpurun_out/r4s2_upload_dbg.txt
timeout 900 python -m pytest tests/test_stream_segments.py tests/test_capture_stream.py -x -q -m gpu > gpurun_out/r4s2_pytest5.txt 2>&1
echo "pytest segments rc=$?"; tail -5 gpurun_out/r4s2_pytest5.txt
timeout 600 python -m pytest tests/test_full_size.py -x -q -m gpu -k "streamed" > gpurun_out/r4s2_pytest5b.txt 2>&1
echo "pytest full size rc=$?"; tail -3 gpurun_out/r4s2_pytest5b.txt
timeout 400 python tools/r4_upload_probe.py 8 4 16 2 12 2>&1 | grep -v amdgpu.ids | grep -v "resident single" > gpurun_out/r4s2_upload.txt
cat gpurun_out/r4s2_upload.txt
