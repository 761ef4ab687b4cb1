#!/bin/bash
# round 5, first GPU call: boundary anatomy of the hot kernel, the RCCL two-ranks-on-one-GPU trial, the gpu suite after the pruning
mkdir -p gpurun_out
timeout 600 python tools/boundary_probe.py > gpurun_out/r05_boundary_anatomy.txt 2> gpurun_out/r05_boundary_anatomy.err
echo "probe rc=$?"; tail -5 gpurun_out/r05_boundary_anatomy.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/rccl_same_gpu_probe.py > gpurun_out/r05_rccl_two_ranks_one_gpu.txt 2>&1
echo "rccl rc=$?"; grep "rank" gpurun_out/r05_rccl_two_ranks_one_gpu.txt | head -12
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05_gpu_pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r05_gpu_pytest.txt
