#!/usr/bin/env python3
"""Where does a sharded pipelined pass spend its HOST time?  1-rank RCCL group on one GPU (the --gpus N code path of bench.py):
per-phase perf_counter sums over K passes.  The wait for the scratch arena of three passes ago sits in runs_begin
(urhgpu_shard_prelaunch_dev -> begin_pipelined_pass): a GPU-bound loop shows it there, a host-bound loop shows none.
env: RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=..."""
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from urh_amd.pipeline import DemodParams
from urh_amd.shard_engine import GpuShardEngine
from urh_amd.sharding import RcclComm, ShardedPipeline, TorchDistComm
from urh_amd.synth import spec_fsk_capture


def main():
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
    iq, _ = spec_fsk_capture(int(os.environ.get("SEGMENTS", "128")), dev, first_segment=0, sps=100)
    e = GpuShardEngine(0, pipelined=True)
    sp = ShardedPipeline(e, TorchDistComm() if os.environ.get("COMM", "rccl") == "torch" else RcclComm.create())
    sp.reserve(iq.shape[0], p)
    acc = {}

    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
            return r
        return w
    for name in ("tail", "runs_begin", "runs", "rows", "bits_prepare", "bits_finish"):
        setattr(e, name, timed(name, getattr(e, name)))
    c = sp.comm
    c.all_gather_start = timed("all_gather_start", c.all_gather_start)
    c.all_gather = timed("all_gather(incl. start)", c.all_gather)
    given = os.environ.get("HALO", "given") == "given"
    for _ in range(300):
        sp.iq_to_bits(iq, p, halo_given=given)
    sp.ctx.join(); torch.cuda.synchronize()
    gc.collect(); gc.disable()
    out = {}
    for K in (100, 400):
        acc.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            sp.iq_to_bits(iq, p, halo_given=given)
        t_issue = time.perf_counter() - t0
        sp.ctx.join(); torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        out[K] = dict(ms_per_step=round(t_all / K * 1e3, 4), issue_ms_per_step=round(t_issue / K * 1e3, 4),
                      host_us_per_step={k: round(v / K * 1e6, 1) for k, v in acc.items()})
    print(json.dumps(dict(comm=type(sp.comm).__name__, halo_given=given, **{str(k): v for k, v in out.items()})))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
