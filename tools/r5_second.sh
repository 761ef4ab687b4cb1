#!/bin/bash
# round 5, second GPU call: the hot kernel's wave-level timeline inside the product's loops; bubbles between hot kernels
mkdir -p gpurun_out
timeout 400 python tools/inrun_anatomy.py > gpurun_out/r05_inrun_anatomy.txt 2> gpurun_out/r05_inrun_anatomy.err
echo "inrun rc=$?"; tail -3 gpurun_out/r05_inrun_anatomy.err
timeout 300 python tools/inrun_anatomy.py --no-stamps > gpurun_out/r05_inrun_nostamps.txt 2>&1
echo "nostamps rc=$?"; grep "==" gpurun_out/r05_inrun_nostamps.txt
timeout 300 python tools/boundary_probe.py --variants B,K,L,M,A > gpurun_out/r05_boundary_bubbles.txt 2>&1
echo "bubbles rc=$?"
