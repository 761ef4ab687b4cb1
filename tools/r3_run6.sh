#!/bin/bash
mkdir -p gpurun_out/r3f
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r3f/tests.txt
timeout 900 python bench.py > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err
tail -2 gpurun_out/r3f/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3f/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print({k: v for k, v in d["config"].items() if k not in ("workload", "capture", "timed_region", "d2h_format", "ranks")})
print(d["roofline"])
print("parity", d.get("parity", {}).get("bit_exact"))
for e in d.get("extra", []):
    print(e.get("workload", "")[:40], e.get("ms"), e.get("stages_ms"), e.get("estimate_stages_ms"), e.get("for_comparison_ms"), (e.get("parity") or {}).get("bit_exact"), e.get("error"), e.get("trace"))
PY
