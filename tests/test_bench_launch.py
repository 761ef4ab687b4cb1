"""bench.py's launch contract on a box without GPUs: `python bench.py --gpus N` (N > 1) from a bare interpreter must start its own
ranks (VERDICT r01 Weak 10: the driver does not wrap the command in torch.distributed.run)."""
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_self_launches_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side launch check; the GPU box runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    err = r.stdout + r.stderr
    assert r.returncode != 0
    # both ranks got as far as the GPU check: the launch itself (rendezvous on 127.0.0.1, RANK / WORLD_SIZE plumbing) worked
    assert err.count("bench.py needs a GPU (no CPU fallback)") >= 2, err[-3000:]
    assert "WORLD_SIZE=1" not in err
