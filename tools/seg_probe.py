#!/usr/bin/env python3
"""Round 4: what a streamed pass looks like on the GPU.  python tools/seg_probe.py <segments> <unused> [singles] [burst]
single captures (push + flush, the machine idle before each), then a burst of back-to-back pushes; run under rocprofv3 --kernel-trace
and read the timeline with tools/seg_timeline.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture

S, shape = int(sys.argv[1]), int(sys.argv[2])
singles = int(sys.argv[3]) if len(sys.argv) > 3 else 4
burst = int(sys.argv[4]) if len(sys.argv) > 4 else 24
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(int(os.environ.get("SEGMENTS", "128")), dev)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
tuning = {"stream_policy": int(os.environ.get("POLICY", "1")), "stream_segments": S}
pipe = DevicePipeline(0, pipelined=True, tuning=tuning)
pipe.reserve(n, p)
st = pipe.stream(n, p, want_qad=True, want_pos=False)
for _ in range(3):
    st.push(iq); st.flush()
torch.cuda.synchronize()
# warm clocks
for _ in range(100):
    st.push(iq)
st.flush(); torch.cuda.synchronize()
one = []
for _ in range(singles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.push(iq); st.flush()
    one.append((time.perf_counter() - t0) * 1e3)
    time.sleep(0.002)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(burst):
    st.push(iq)
st.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / burst * 1e3
sys.stdout.flush()
print(f"segments {S} shape {shape} bits segments {os.environ.get('SB', '3')}: single capture {min(one):.4f} ms (all {[round(x, 4) for x in one]}), burst {dt:.4f} ms per pass, stats {st.stats()}")
st.close()
sys.stdout.flush()
