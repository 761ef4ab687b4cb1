#!/usr/bin/env python3
"""Steady-state and short-run step times of the capture stream (urhgpu_stream_*): with / without bit_sample_pos, in both creation orders,
for K = 20 and K = 100 steps per timed region (the K steps include the drain: last tail, pack and copy)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dataclasses import replace
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture

dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
pipe = DevicePipeline(0, pipelined=True)
pipe.reserve(n, p)


KEEP = os.environ.get("PROBE_KEEP") == "1"


def run(st, k):
    out = []
    for _ in range(k):
        r = st.push(iq)
        if KEEP and r is not None:
            out.append(r)
    return out + st.flush()


def measure(want_pos, tag):
    st = pipe.stream(n, replace(p, write_bit_sample_pos=want_pos), want_qad=True, want_pos=want_pos)
    for _ in range(12):
        run(st, 10)
    out = []
    for k in (20, 100, 20, 100):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = run(st, k)
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / k * 1e3, 4))
    print(tag, "want_pos", want_pos, "ms/step K=20,100,20,100:", out, "blob bytes", r[-1].blob_bytes, flush=True)
    st.close()


def device_only(k):
    for _ in range(100):
        pipe.iq_to_bits(iq, p, want_qad=True)
    pipe.ctx.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        pipe.iq_to_bits(iq, p, want_qad=True)
    pipe.ctx.join(); torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / k * 1e3, 4)


print("device only K=20, 100:", device_only(20), device_only(100), flush=True)
measure(False, "first ")
measure(True, "second")
measure(False, "third ")
measure(True, "fourth")
print("device only K=20, 100:", device_only(20), device_only(100), flush=True)
