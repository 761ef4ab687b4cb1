#!/bin/bash
mkdir -p gpurun_out/r3j
python -m pytest tests/test_capture_stream.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r3j/tests.txt
bash tools/r3_ab.sh gpurun_out/r3j/ab.txt 2 default default:URH_TAIL_RESERVED_CUS=16 default:URH_TAIL_RESERVED_CUS=32
URH_TAIL_RESERVED_CUS=16 bash tools/r3_prof.sh r3j/prof_res16
