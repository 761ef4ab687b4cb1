"""Step time over consecutive groups of 10 passes from a cold start (clock / allocation ramp): developer probe"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
for pipelined in (True, False):
    pipe = DevicePipeline(0, pipelined=pipelined)
    pipe.reserve(iq.shape[0], p)
    for i in range(3): r = pipe.iq_to_bits(iq, p, want_qad=True)
    pipe.ctx.join(); torch.cuda.synchronize()
    out = []
    for g in range(12):
        t0 = time.perf_counter()
        for i in range(10): r = pipe.iq_to_bits(iq, p, want_qad=True)
        pipe.ctx.join(); torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / 10 * 1e3, 4))
    print("pipelined" if pipelined else "one after the other", out)
    time.sleep(0.5)
    out = []
    for g in range(4):
        t0 = time.perf_counter()
        for i in range(10): r = pipe.iq_to_bits(iq, p, want_qad=True)
        pipe.ctx.join(); torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / 10 * 1e3, 4))
    print("  after 0.5 s idle", out)
    del pipe
