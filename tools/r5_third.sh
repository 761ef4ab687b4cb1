#!/bin/bash
# round 5, third GPU call: what slows the hot kernel inside the run?  tail kernels left out one by one; synthetic company; tail priority
mkdir -p gpurun_out
timeout 500 python tools/inrun_anatomy.py --skips 0,63,62,2,16,18,12,32,1,61,0 > gpurun_out/r05_tail_skips.txt 2> gpurun_out/r05_tail_skips.err
echo "skips rc=$?"; tail -3 gpurun_out/r05_tail_skips.err
timeout 400 python tools/boundary_probe.py --variants C,N,O,P,Q,R,C > gpurun_out/r05_company.txt 2>&1
echo "company rc=$?"
URHGPU_LIB=$PWD/urh_amd/liburhgpu_prio0.so timeout 300 python tools/inrun_anatomy.py --skips 0 > gpurun_out/r05_tail_prio0.txt 2>&1
echo "prio0 rc=$?"; grep -A3 "==" gpurun_out/r05_tail_prio0.txt | head
