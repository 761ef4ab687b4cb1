"""Sharded pipelined passes alternating between TWO engines (contexts, tail streams, process groups): every tail then has two hot-kernel
periods to finish.  Developer probe, 1-rank RCCL groups:
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29516 tools/sharded_two_engines.py"""
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
n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pipes = [ShardedPipeline(GpuShardEngine(0, pipelined=True), TorchDistComm(dist.new_group(backend="nccl") if k else None))
         for k in range(n_eng)]


def run(k_passes):
    for i in range(k_passes):
        r = pipes[i % n_eng].iq_to_bits(iq, p, want_qad=True)
    return r


def wait_all():
    for sp in pipes:
        sp.ctx.join()
    torch.cuda.synchronize()


run(150); wait_all()
for rep in range(4):
    t0 = time.perf_counter()
    r = run(40)
    t1 = time.perf_counter()
    wait_all()
    t2 = time.perf_counter()
    print(f"{n_eng} engine(s): enqueue {1e3 * (t1 - t0) / 40:.4f} ms per pass, done after {1e3 * (t2 - t0) / 40:.4f} ms per pass", r.host_counts())
dist.destroy_process_group()
