#!/bin/bash
mkdir -p gpurun_out/r3y
python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_reference_dropin.py tests/test_signal_shim.py -m gpu -q -x -k "center or estimate or statistic or config3 or config5 or detect or hot_path or signal" 2>&1 | tail -3 | tee gpurun_out/r3y/tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d.get('extra', []):
    print(e.get('workload', '')[:30], e.get('ms'), e.get('stages_ms'), e.get('estimate_stages_ms'), e.get('error'))
" | tee gpurun_out/r3y/extras.txt
