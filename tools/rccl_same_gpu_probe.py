#!/usr/bin/env python3
"""Can RcclComm (urh_amd/sharding.py) run with MORE THAN ONE rank on a 1-GPU box -- two processes, the same device?  

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/rccl_same_gpu_probe.py

torch.distributed runs over gloo here (its nccl backend refuses two ranks on one device on its own account); the library's communicator
is made exactly as in production (ncclGetUniqueId on rank 0, the id through the group, ncclCommInitRank bounded by a timeout) with both
ranks on cuda:0.  Prints RCCL's verdict -- the error string goes into DESIGN.md section 5 -- and, should it come up, runs the sharded
IQ->bits pass on two shards and checks the stitched result against a single-GPU pass."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist
    from urh_amd.sharding import RcclComm
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    RcclComm.INIT_TIMEOUT_S = 60.0
    comm, reason = RcclComm._create(None, RcclComm.load_library, torch.device("cpu"))
    print(f"[rank {rank}] RcclComm with {world} ranks on ONE device: {'UP' if comm is not None else 'refused'}; reason: {reason}", flush=True)
    if comm is None:
        dist.barrier()
        dist.destroy_process_group()
        return
    try:
        t = torch.full((4,), float(rank + 1), device=dev)
        out = comm.all_gather(t)
        torch.cuda.synchronize()
        print(f"[rank {rank}] all_gather -> {out.cpu().numpy().tolist()}", flush=True)
        from conftest import synth_fsk
        from urh_amd.pipeline import DemodParams, DevicePipeline
        from urh_amd.shard_engine import GpuShardEngine
        from urh_amd.sharding import ShardedPipeline, shard_bounds, stitch
        n = (4 << 20)
        iq = synth_fsk(n, sps=100, seed=7, noise=0.05, pause_every=700_000, pause_len=40_000)
        a, b = shard_bounds(n, world)[rank]
        shard = torch.from_numpy(iq[a:b]).to(dev)
        p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
        for pipelined in (False, True):
            pipe = ShardedPipeline(GpuShardEngine(0, pipelined=pipelined), comm)
            for _ in range(3):
                res = pipe.iq_to_bits(shard, p, want_qad=True, pos_base=a, n_total=n)
            pipe.ctx.join()
            torch.cuda.synchronize()
            pieces = [None] * world
            dist.all_gather_object(pieces, res.piece())
            if rank == 0:
                single = DevicePipeline(0).iq_to_bits(torch.from_numpy(iq).to(dev), p, want_qad=False)
                got = stitch(pieces)
                ok = np.array_equal(got[0], single.ppseq()) and all(np.array_equal(x, y) for x, y in zip(got[1:], single.flat()))
                print(f"[rank 0] two ranks over RcclComm on one GPU, pipelined={pipelined}: stitched result equals the single-GPU pass: {ok}", flush=True)
            if pipelined:
                pipe.ctx.set_pipelined(False)
    except Exception as exc:                                   # noqa: BLE001
        print(f"[rank {rank}] after the communicator came up: {exc!r}", flush=True)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
