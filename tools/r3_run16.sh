#!/bin/bash
mkdir -p gpurun_out/r3v
python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_reference_dropin.py -m gpu -q -x -k "center or estimate or statistic or config3 or config5 or detect or hot_path" 2>&1 | tail -3 | tee gpurun_out/r3v/tests.txt
PYTHONPATH=. python tools/prof_estimate.py --reps 20 2>&1 | tail -1 | tee gpurun_out/r3v/estimate.txt
PYTHONPATH=. python tools/center_only.py 2>&1 | tail -3 | tee -a gpurun_out/r3v/estimate.txt
