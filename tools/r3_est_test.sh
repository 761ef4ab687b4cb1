#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "segmentation_pass or estimate or segment or auto_interp or full_size or dropin" 2>&1 | tail -4
python - <<'PY'
import subprocess, json, sys
out = subprocess.run([sys.executable, "tools/est_probe.py", "--no-psk"], capture_output=True, text=True)
for l in out.stdout.splitlines():
    if l.startswith("{"): print(l[:900])
print(out.stderr[-600:] if out.returncode else "")
PY
