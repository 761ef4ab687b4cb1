#!/bin/bash
# round-3 GPU check 1: new tests + bench line
mkdir -p gpurun_out/r3a
python -m pytest tests/test_capture_stream.py tests/test_signal_shim.py tests/test_auto_interpretation_dropin.py tests/test_reference_dropin.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r3a/tests.txt
cat gpurun_out/r3a/tests.txt
timeout 900 python bench.py --no-extra > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -3 gpurun_out/r3a/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3a/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"])
    print({k: v for k, v in d["config"].items() if "ms" in k or "d2h" in k or "bytes" in k or "parity" in k})
    print(d["roofline"])
    print(d.get("parity"))
    print({k: d["cpu_baseline"][k] for k in ("value", "cores", "kind")})
except Exception as e:
    print("bench parse failed", e)
PY
