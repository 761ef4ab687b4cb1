#!/bin/bash
# round 5, fourth GPU call: the fused row stage -- its own tests first, then what it does to the loops
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_rows.py -x -q > gpurun_out/r05_fused_tests.txt 2>&1
echo "fused tests rc=$?"; tail -15 gpurun_out/r05_fused_tests.txt
timeout 400 python tools/inrun_anatomy.py > gpurun_out/r05_fused_inrun.txt 2> gpurun_out/r05_fused_inrun.err
echo "inrun rc=$?"; tail -3 gpurun_out/r05_fused_inrun.err; grep -A4 "^==" gpurun_out/r05_fused_inrun.txt | cut -c1-200
URH_TUNE_HOT_FUSED_ROWS=0 timeout 400 python tools/inrun_anatomy.py > gpurun_out/r05_unfused_inrun.txt 2>&1
echo "unfused rc=$?"; grep -A4 "^==" gpurun_out/r05_unfused_inrun.txt | cut -c1-200
