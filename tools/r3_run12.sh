#!/bin/bash
mkdir -p gpurun_out/r3p
python -m pytest tests/test_capture_stream.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r3p/tests.txt
for k in 20 20 100; do
python bench.py --no-extra --steps $k --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('K=$k value', d['value'], 'ms', d['ms_per_step'], 'withpos', c.get('ms_per_step_with_device_positions'), c.get('device_positions_equal_derived'), 'dev', c['device_only_ms_per_step'], 'k', r['kernel_ms'], r['frac'], 'e2e', r['end_to_end_frac'], r['end_to_end_device_only_frac'], 'parity', c.get('parity_bit_exact'), 'd2h', c.get('d2h_bytes_per_step'), 'ceil', r['copy_ceiling_gbs'])" | tee -a gpurun_out/r3p/bench.txt
done
