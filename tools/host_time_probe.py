"""Host-side enqueue time of one pass (no GPU sync inside the loop): single-GPU pipeline vs the sharded path with a 1-rank group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
import torch, torch.distributed as dist
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.shard_engine import GpuShardEngine
from urh_amd.sharding import ShardedPipeline, TorchDistComm
from urh_amd.synth import fsk_capture
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl")
iq, _ = fsk_capture(128, dev, seed=1234)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
for name, pipe in (("single", DevicePipeline(0)), ("sharded", ShardedPipeline(GpuShardEngine(0), TorchDistComm()))):
    pipe.reserve(iq.shape[0], p)
    for _ in range(3): pipe.iq_to_bits(iq, p, want_qad=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): pipe.iq_to_bits(iq, p, want_qad=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3 * (t1 - t0) / 50:.3f} ms/step, wall {1e3 * (t2 - t0) / 50:.3f} ms/step")
dist.destroy_process_group()
