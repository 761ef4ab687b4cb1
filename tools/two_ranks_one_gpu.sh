#!/bin/bash
# The N > 1 bench line with N REAL PROCESSES on a 1-GPU box: every rank on device 0, the group over gloo (RCCL refuses two ranks on one
# device: profiles/r05_rccl_two_ranks_one_gpu.txt), the exchanges through TorchDistComm's host path.  What it proves: the rank > 0
# branches of bench.py and of the sharded protocol (halo handed over with the shard, summaries of earlier AND later ranks, flags, the
# FIR-halo variant with its raw halo, stitch, the self-check against ONE single-GPU pass over the N-GiB capture and against oracle/_ref)
# run as separate processes and agree bit for bit -- at N = 8 that is BASELINE.json configs[3]'s capture.  Its timings mean nothing
# (the ranks share the GPU, the exchanges cross the host).
#   tools/two_ranks_one_gpu.sh            selftests at N = 2, 4, 8; the whole bench line (FSK + FIR-halo variant) at N = 2 and 8
#   -> gpurun_out/ranks<N>_one_gpu_{selftest,bench}.json and a summary on stdout
set -u
mkdir -p gpurun_out
export URH_BENCH_SHARE_GPU=1 URH_BENCH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
port=29540
run() {     # N name args...
    local N=$1 name=$2; shift 2
    port=$((port + 1))
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$port" \
        bench.py --gpus "$N" "$@" > "gpurun_out/$name.json" 2> "gpurun_out/$name.err"
    local rc=$?
    python - "$name" "$rc" <<'PY'
import json, sys
name, rc = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/{name}.json").read().strip().splitlines()[-1])
    c = d["config"]
    sp = c.get("sharded_parity") or {}
    fh = c.get("fir_halo") or {}
    print(f"{name}: rc={rc} n_gpus={d['n_gpus']} collectives={c.get('collectives')} parity_bit_exact={c.get('parity_bit_exact')} "
          f"rows={sp.get('rows')} bits={sp.get('n_bits')} oracle_shard0={(sp.get('oracle_shard0') or {}).get('bit_exact')}"
          + (f" | fir_halo bit_exact={fh.get('bit_exact')} rows={fh.get('rows')} bits={fh.get('n_bits')}" if fh else ""))
except Exception as exc:
    print(f"{name}: rc={rc} NO LINE ({exc!r})")
PY
}
for N in 2 4 8; do run $N ranks${N}_one_gpu_selftest --selftest-only; done
for N in 2 8; do run $N ranks${N}_one_gpu_bench --steps 3 --warmup 1 --no-variants; done
