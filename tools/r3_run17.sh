#!/bin/bash
mkdir -p gpurun_out/r3w
python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_reference_dropin.py tests/test_signal_shim.py -m gpu -q -x -k "center or estimate or statistic or config3 or config5 or detect or hot_path or signal" 2>&1 | tail -3 | tee gpurun_out/r3w/tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r3w/bench.json 2> gpurun_out/r3w/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3w/bench.json").read().strip().splitlines()[-1])
c = d["config"]; r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "withpos", c.get("ms_per_step_with_device_positions"), "dev", c["device_only_ms_per_step"], "k", r["kernel_ms"], r["frac"], "e2e", r["end_to_end_frac"], "parity", c.get("parity_bit_exact"))
for e in d.get("extra", []):
    print(e.get("workload", "")[:30], e.get("ms"), e.get("stages_ms"), e.get("estimate_stages_ms"), (e.get("parity") or {}).get("bit_exact"), e.get("error"))
PY
