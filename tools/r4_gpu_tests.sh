#!/bin/bash
# the whole -m gpu suite (as the driver runs it) + the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4_gpu_pytest.txt 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r4_gpu_pytest.txt
if [ "$1" != "nobench" ]; then
  timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err
  echo "bench rc=$?"; tail -3 gpurun_out/r4_bench.err
  python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4_bench.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("ms/step", d["ms_per_step"], "value", d["value"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "e2e", d["roofline"].get("end_to_end_frac"))
    for k in ("single_capture_incl_compact_d2h_ms", "ms_per_step_with_device_positions", "device_only_ms_per_step", "unpipelined_ms_per_step", "parity_bit_exact", "configs2_ook_fir", "configs4_psk_costas", "stream_stats"):
        print(k, c.get(k))
    for ex in d.get("extra", []):
        print(ex.get("workload", "")[:40], ex.get("ms"), ex.get("stages_ms"), (ex.get("parity") or {}).get("bit_exact"), ex.get("error"))
    print("parity", d.get("parity"))
except Exception as e:
    print("no bench line:", e)
PY
fi
