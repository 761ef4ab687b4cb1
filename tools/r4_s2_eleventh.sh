#!/bin/bash
# the whole -m gpu suite, the default bench line, the forced-sharded (1-rank RCCL group) line with its self-check and FIR-halo record
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4_gpu_pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r4_gpu_pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r4_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4_bench.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("ms/step", d["ms_per_step"], "value", d["value"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "e2e", d["roofline"].get("end_to_end_frac"))
    for k in ("single_capture_incl_compact_d2h_ms", "ms_per_step_with_device_positions", "device_only_ms_per_step", "unpipelined_ms_per_step", "parity_bit_exact",
              "bare_pinned_h2d_ms", "h2d_inclusive_ms", "h2d_inclusive_over_bare", "h2d_inclusive_equals_resident_result", "h2d_inclusive_error", "configs2_ook_fir", "configs4_psk_costas", "stream_stats"):
        print(k, c.get(k))
    for ex in d.get("extra", []):
        print(ex.get("workload", "")[:40], ex.get("ms"), ex.get("stages_ms"), (ex.get("parity") or {}).get("bit_exact"), ex.get("error"))
except Exception as e:
    print("no bench line:", e)
PY
URH_BENCH_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_sharded1.json 2> gpurun_out/r4_bench_sharded1.err
echo "sharded bench rc=$?"; tail -3 gpurun_out/r4_bench_sharded1.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4_bench_sharded1.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("sharded ms/step", d["ms_per_step"], "device only", c.get("device_only_ms_per_step"), "collectives", c.get("collectives"), c.get("collectives_fallback_reason"))
    print("sharded_parity", c.get("sharded_parity"))
    print("fir_halo", c.get("fir_halo"))
except Exception as e:
    print("no sharded line:", e)
PY
