#!/usr/bin/env python3
"""Round 4: latency of ONE streamed capture (push + flush on an idle GPU) and the burst step for a grid of segmentations, one process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture

dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev)
n = iq.shape[0]
want_pos = bool(int(os.environ.get("WANT_POS", "0")))
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, want_pos)
grid = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1, 1, 2), (4, 2, 1), (6, 3, 1), (8, 3, 1)]
for S, Sb, policy in grid:
    pipe = DevicePipeline(0, pipelined=True, tuning={"stream_policy": policy, "stream_segments": S, "stream_bits_segments": Sb})
    pipe.reserve(n, p)
    st = pipe.stream(n, p, want_qad=True, want_pos=want_pos)
    for _ in range(3):
        st.push(iq); st.flush()
    for _ in range(150):                     # clocks
        st.push(iq)
    st.flush(); torch.cuda.synchronize()
    one = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st.push(iq); st.flush()
        one.append((time.perf_counter() - t0) * 1e3)
    for _ in range(100):
        st.push(iq)
    st.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        st.push(iq)
    st.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40 * 1e3
    one.sort()
    print(f"rows segments {S:2d} bits segments {Sb:2d} policy {policy}: one capture min {one[0]:.4f} median {one[len(one) // 2]:.4f} ms; burst of 40: {dt:.4f} ms per pass; {st.stats()}", flush=True)
    st.close()
    del st, pipe
