#!/bin/bash
# round 5, seventh GPU call: mask policy for integer captures (A/B), dynamic instruction counts per sample type, the FIR-halo variant with the
# raw halo handed over (test + the 1-rank sharded line), --selftest-only
mkdir -p gpurun_out
rm -f gpurun_out/r05_dtype_stream_ab.txt
for lib in liburhgpu.so liburhgpu_maskall.so; do
  URHGPU_LIB=$PWD/urh_amd/$lib timeout 300 python tools/dtype_stream_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_dtype_stream_ab.txt
done
cat gpurun_out/r05_dtype_stream_ab.txt
bash tools/dtype_pmc.sh gpurun_out/r05_dtype_pmc.txt; cat gpurun_out/r05_dtype_pmc.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_dropin.py -x -q -m gpu -k "sharded or rccl" > gpurun_out/r05_shard_tests.txt 2>&1
echo "shard tests rc=$?"; tail -4 gpurun_out/r05_shard_tests.txt
URH_BENCH_FORCE_SHARDED=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --selftest-only > gpurun_out/r05_selftest_1rank.json 2> gpurun_out/r05_selftest_1rank.err
echo "selftest rc=$?"; cut -c1-600 gpurun_out/r05_selftest_1rank.json
URH_BENCH_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_sharded_1rank.json 2> gpurun_out/r05_bench_sharded_1rank.err
echo "sharded bench rc=$?"; tail -2 gpurun_out/r05_bench_sharded_1rank.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench_sharded_1rank.json").read().strip().splitlines()[-1]); c = d["config"]
    print("ms/step", d["ms_per_step"], "device-only", c.get("device_only_ms_per_step"), "parity", c.get("parity_bit_exact"))
    print("fir_halo", json.dumps(c.get("fir_halo"))[:1400])
except Exception as e:
    print("no line:", e)
PY
