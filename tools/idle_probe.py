#!/usr/bin/env python3
"""How sensitive is the step time of K = 20 pipelined sharded passes to a short idle period of the GPU right before them?
(1-rank RCCL group.)  For each idle time: 200 passes, drain, [dist.barrier()], sleep, 20 timed passes."""
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
from urh_amd.sharding import RcclComm, ShardedPipeline
from urh_amd.synth import spec_fsk_capture


def main():
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
    iq, _ = spec_fsk_capture(128, dev, first_segment=0, sps=100)
    sp = ShardedPipeline(GpuShardEngine(0, pipelined=True), RcclComm.create())
    sp.reserve(iq.shape[0], p)

    def steps(k):
        for _ in range(k):
            sp.iq_to_bits(iq, p, halo_given=True)
        sp.ctx.join()
    steps(300)
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    out = []
    for barrier in (False, True):
        for idle_ms in (0.0, 0.2, 1.0, 5.0, 20.0, 100.0):
            rec = []
            for rep in range(3):
                steps(200)
                torch.cuda.synchronize()
                tb = time.perf_counter()
                if barrier:
                    dist.barrier()
                    torch.cuda.synchronize()
                tb = time.perf_counter() - tb
                if idle_ms:
                    time.sleep(idle_ms * 1e-3)
                t0 = time.perf_counter()
                steps(20)
                torch.cuda.synchronize()
                rec.append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
            out.append(dict(barrier=barrier, idle_ms=idle_ms, barrier_ms=round(tb * 1e3, 3), ms_per_step_k20=rec))
    t0 = time.perf_counter(); steps(400); torch.cuda.synchronize()
    out.append(dict(k400=round((time.perf_counter() - t0) / 400 * 1e3, 4)))
    for o in out:
        print(json.dumps(o))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
