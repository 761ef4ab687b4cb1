#!/bin/bash
mkdir -p gpurun_out/r3q
python -m pytest tests/test_capture_stream.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r3q/tests.txt
for k in 100 100 20; do
python bench.py --no-extra --no-cpu-baseline --steps $k --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('K=$k value', d['value'], 'ms', d['ms_per_step'], 'withpos', c.get('ms_per_step_with_device_positions'), c.get('with_device_positions_stream_stats'), c.get('stream_stats'), 'dev', c['device_only_ms_per_step'], 'k', r['kernel_ms'])" | tee -a gpurun_out/r3q/bench.txt
done
