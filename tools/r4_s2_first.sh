#!/bin/bash
# round 4, session 2, first GPU call: the whole -m gpu suite, the default bench line, the segmentation sweep
mkdir -p gpurun_out
bash tools/r4_gpu_tests.sh > gpurun_out/r4s2_tests_bench.txt 2>&1
cat gpurun_out/r4s2_tests_bench.txt
timeout 400 python tools/r4_sweep.py 1,1,2 4,2,1 6,3,1 8,3,1 8,3,0 12,3,1 > gpurun_out/r4s2_sweep.txt 2>&1
cat gpurun_out/r4s2_sweep.txt
WANT_POS=1 timeout 300 python tools/r4_sweep.py 1,1,2 8,3,1 8,3,0 > gpurun_out/r4s2_sweep_pos.txt 2>&1
cat gpurun_out/r4s2_sweep_pos.txt
