"""The N > 1 bench line executed by REAL processes on the one GPU of the test box: every rank on device 0, the process group over gloo
(RCCL refuses two ranks on one device -- profiles/r05_rccl_two_ranks_one_gpu.txt -- so the exchanges take TorchDistComm's host path).
`bench.py --selftest-only` runs one sharded pass and checks the stitched pieces against ONE single-GPU pass over the whole capture and
rank 0's shard against oracle/_ref (signal_functions.pyx:333-495 across shard boundaries)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_selftest_with_real_ranks_on_one_gpu(world):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(URH_BENCH_SHARE_GPU="1", URH_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29570 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--selftest-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world
    sp = line["config"]["sharded_parity"]
    assert line["config"]["parity_bit_exact"] is True and sp["bit_exact"] is True, sp
    for k in ("rows_equal", "bits_equal", "msg_off_equal", "pauses_equal", "bit_sample_pos_equal", "pos_off_equal"):
        assert sp[k] is True, (k, sp)
    assert sp["qad_rank0_shard_mismatches"] == 0 and sp["oracle_shard0"]["bit_exact"] is True
