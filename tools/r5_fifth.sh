#!/bin/bash
# round 5, fifth GPU call: integer kernels held to seven wavefronts per SIMD (A/B), the gpu suite, the bench line with its new keys
mkdir -p gpurun_out
for lib in liburhgpu.so liburhgpu_intw6.so; do
  echo "== $lib" >> gpurun_out/r05_dtypes_ab.txt
  URHGPU_LIB=$PWD/urh_amd/$lib timeout 300 python tools/dtype_probe.py >> gpurun_out/r05_dtypes_ab.txt 2>&1
done
cat gpurun_out/r05_dtypes_ab.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05_gpu_pytest2.txt 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r05_gpu_pytest2.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_a.json 2> gpurun_out/r05_bench_a.err
echo "bench rc=$?"; tail -3 gpurun_out/r05_bench_a.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench_a.json").read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
    print("ms/step", d["ms_per_step"], "kernel", r["kernel_ms"], "frac", r["frac"], "e2e", r.get("end_to_end_frac"))
    print("variants", json.dumps(c.get("variants"))[:1500])
    print("with pos", c.get("ms_per_step_with_device_positions"), c.get("value_with_positions"), "single", c.get("single_capture_incl_compact_d2h_ms"), "parity", c.get("parity_bit_exact"))
    print(c.get("configs2_ook_fir"), c.get("configs4_psk_costas"))
except Exception as e:
    print("no bench line:", e)
PY
