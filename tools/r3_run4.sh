#!/bin/bash
mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_parity.py tests/test_capture_stream.py -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/r3d/tests.txt
bash tools/r3_ab.sh gpurun_out/r3d/ab.txt 2 default noprio cpw1 cpw4 cpw8 w8 default:URH_TAIL_STREAM_PRIORITY=-1 cpw1:URH_TAIL_STREAM_PRIORITY=-1
bash tools/r3_prof.sh r3d/prof_default
URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_cpw1.so bash tools/r3_prof.sh r3d/prof_cpw1
