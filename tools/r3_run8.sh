#!/bin/bash
mkdir -p gpurun_out/r3h
python -m pytest tests/test_capture_stream.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r3h/tests.txt
bash tools/r3_ab.sh gpurun_out/r3h/ab.txt 2 default cpw8 default:URH_HOT_LDS_KB=24
