"""Is the sharded pass host-bound?  Host time to ENQUEUE K passes against the time until they are done (1-rank RCCL group): developer probe
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29514 tools/sharded_host_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from urh_amd.pipeline import DemodParams
from urh_amd.shard_engine import GpuShardEngine
from urh_amd.sharding import ShardedPipeline, TorchDistComm
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
iq, _ = spec_fsk_capture(128, dev)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
pipe = ShardedPipeline(GpuShardEngine(0, pipelined=True), TorchDistComm())
for _ in range(100): r = pipe.iq_to_bits(iq, p, want_qad=True)
pipe.ctx.join(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(40): r = pipe.iq_to_bits(iq, p, want_qad=True)
    t1 = time.perf_counter()
    pipe.ctx.join(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / 40:.4f} ms per pass, done after {1e3 * (t2 - t0) / 40:.4f} ms per pass")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(40): r = pipe.iq_to_bits(iq, p, want_qad=True)
pr.disable(); pipe.ctx.join(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
dist.destroy_process_group()
