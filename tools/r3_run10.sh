#!/bin/bash
mkdir -p gpurun_out/r3l
python -m pytest tests/test_capture_stream.py tests/test_abi.py tests/test_gpu_parity.py -m gpu -q -x -k "stream or abi or pipelined or sharded" 2>&1 | tail -3 | tee gpurun_out/r3l/tests.txt
bash tools/r3_ab.sh gpurun_out/r3l/ab.txt 2 default default:URH_HOT_CUS_REMOVED=0 default:URH_HOT_CUS_REMOVED=3 default:URH_HOT_CUS_REMOVED=6
bash tools/r3_prof.sh r3l/prof_default
