"""Where does a sharded pipelined pass spend its time on the GPU?  rocprofv3 slows the host below the GPU's pace on this path, so the
phases are bracketed with timing events instead (tail stream: one event after every phase; main stream: around the hot launch) and
printed as a timeline of a few consecutive passes.  Developer probe, 1-rank RCCL group:
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29515 tools/sharded_timeline.py"""
import os, sys
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
e, c = pipe.engine, pipe.comm
n = int(iq.shape[0])


def ev():
    x = torch.cuda.Event(enable_timing=True)
    x.record()                      # on torch's current stream
    return x


def one_pass(marks):
    """ShardedPipeline.iq_to_bits with an event after every phase"""
    marks.append(("main: pass begins", ev()))
    pending = c.all_gather_start(e.tail(iq, p))
    e.runs_begin(iq, 0, n, 0, 1, p, True)
    marks.append(("main: hot kernel done", ev()))
    with e.tail_context():
        halos = pending()
        summary = e.runs(iq, None, 0, n, 0, 1, p, True)
        marks.append(("tail: halo + local resolve done", ev()))
        s_all = c.all_gather(summary)
        marks.append(("tail: summaries gathered", ev()))
        merge = e.rows(s_all)
        marks.append(("tail: rows done", ev()))
        flags = e.bits_prepare(None)
        marks.append(("tail: bits prepared", ev()))
        f_all = c.all_gather(flags)
        marks.append(("tail: flags gathered", ev()))
        r = e.bits_finish(f_all)
        marks.append(("tail: bits done", ev()))
    return r


for _ in range(120):
    pipe.iq_to_bits(iq, p, want_qad=True)
passes = []
for _ in range(8):
    m = []
    one_pass(m)
    passes.append(m)
pipe.ctx.join(); torch.cuda.synchronize()
t0 = passes[2][0][1]
for k, m in enumerate(passes[2:7]):
    for name, x in m:
        print(f"pass {k}  {t0.elapsed_time(x) * 1e3:9.1f} us  {name}")
dist.destroy_process_group()
