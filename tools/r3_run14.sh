#!/bin/bash
mkdir -p gpurun_out/r3s
python tools/cumask_probe.py 2>&1 | grep "CUs used" | tee gpurun_out/r3s/cumask.txt
bash tools/r3_ab.sh gpurun_out/r3s/ab.txt 3 default default:URH_HOT_CUS_REMOVED=0
python tools/mask_policy_probe.py 2>&1 | grep "ms/step" | head -1 | tee -a gpurun_out/r3s/cumask.txt
