#!/bin/bash
# round 5, sixth GPU call: the hot stream's CU mask for integer captures (A/B), the chained estimate call (tests + timing)
mkdir -p gpurun_out
for lib in liburhgpu.so liburhgpu_maskall.so; do
  URHGPU_LIB=$PWD/urh_amd/$lib timeout 300 python tools/dtype_stream_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_dtype_stream_ab.txt
done
cat gpurun_out/r05_dtype_stream_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "center or estimate or plateau or message or noise" > gpurun_out/r05_est_tests.txt 2>&1
echo "est tests rc=$?"; tail -4 gpurun_out/r05_est_tests.txt
timeout 600 python -m pytest tests/test_full_size.py tests/test_signal_shim.py -x -q -m gpu > gpurun_out/r05_est_tests2.txt 2>&1
echo "full-size rc=$?"; tail -4 gpurun_out/r05_est_tests2.txt
bash tools/est_prof.sh r05a_est > /dev/null 2>&1
for part in ook psk; do echo "== $part"; tail -1 gpurun_out/r05a_est/log_$part.txt | cut -c1-700; head -12 gpurun_out/r05a_est/kernels_$part.txt; done
