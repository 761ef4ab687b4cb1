#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stream_segments.py tests/test_capture_stream.py -x -q -m gpu > gpurun_out/r4s2_pytest3.txt 2>&1
echo "pytest segments rc=$?"; tail -8 gpurun_out/r4s2_pytest3.txt
timeout 600 python -m pytest tests/test_full_size.py -x -q -m gpu -k streamed > gpurun_out/r4s2_pytest3b.txt 2>&1
echo "pytest full size rc=$?"; tail -8 gpurun_out/r4s2_pytest3b.txt
timeout 400 python tools/r4_upload_probe.py > gpurun_out/r4s2_upload.txt 2>&1
cat gpurun_out/r4s2_upload.txt
