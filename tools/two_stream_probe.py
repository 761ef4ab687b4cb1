"""Pipelined passes issued alternately on two streams (the hot kernel of pass i + 1 may start while pass i's drains): developer probe"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
for mode in ("one stream, one slot", "one stream", "two streams"):
    pipe = DevicePipeline(0, pipelined=True)
    pipe.reserve(iq.shape[0], p)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    def step(i):
        if mode == "one stream, one slot":
            return pipe.iq_to_bits(iq, p, want_qad=True, slot=0)
        if mode == "one stream":
            return pipe.iq_to_bits(iq, p, want_qad=True, slot=i & 1)
        with torch.cuda.stream(streams[i & 1]):
            return pipe.iq_to_bits(iq, p, want_qad=True, slot=i & 1)
    for i in range(6): r = step(i)
    pipe.ctx.join(); torch.cuda.synchronize()
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(40): r = step(i)
        pipe.ctx.join(); torch.cuda.synchronize()
        print(mode, round((time.perf_counter() - t0) / 40 * 1e3, 4), r.host_counts())
    del pipe
